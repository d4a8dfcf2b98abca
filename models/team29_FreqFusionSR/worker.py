"""Helper rank of a self-launched multi-GPU ``main`` (FFSR_GPUS=N): ``python worker.py model_dir input_path output_path``
with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment (set by io._self_launch).  The counterpart of the
reference's per-GPU worker script (scripts/kaggle_inference_fixed.py:100-130: ``WORKER_SCRIPT rank num_gpus args``)."""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

if __name__ == "__main__":
    import torch
    from models.team29_FreqFusionSR import main
    main(model_dir=sys.argv[1], input_path=sys.argv[2], output_path=sys.argv[3], device=torch.device("cuda"))
