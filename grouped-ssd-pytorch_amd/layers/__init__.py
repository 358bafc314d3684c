from .functions import *   # noqa: F401,F403
from .modules import *     # noqa: F401,F403
