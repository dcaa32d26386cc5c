from .timers import Timer, Timers  # noqa: F401
from .meters import AverageMeter, Logger, setup_seed  # noqa: F401
