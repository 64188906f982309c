"""PPG time-stretching (reference ppgs/edit/grid.py)."""
from . import grid    # noqa: F401
