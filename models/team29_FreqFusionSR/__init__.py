from .io import *  # noqa: F401,F403  (same export convention as the reference package)
from .io import main  # noqa: F401
