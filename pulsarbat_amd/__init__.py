"""pulsarbat_amd: MI355X-native coherent dedispersion behind the pulsarbat API.

Same public names as the reference package for the hot path
(pulsarbat/__init__.py:9-33): signal containers, ``DispersionMeasure``/``DM``,
``coherent_dedispersion`` and the ``fft`` dispatch module.  ``units`` and
``Time`` stand in for astropy (absent on the target boxes).
"""

__version__ = "0.1.0"

from . import units
from .time import Time
from . import core
from .core import *
from .core import InvalidSignalError
from .device import DeviceArray
from . import transforms
from .transforms import *
from . import fft
from . import contrib
from . import utils
from . import readers
from . import shard     # channel sharding across the GPUs of a node (one process per GPU)
from . import node      # node-level buffers and the peer-write gather behind shard

__all__ = ["fft", "contrib", "utils", "readers", "shard", "node", "units", "Time", "DeviceArray", "InvalidSignalError"]
__all__.extend(core.__all__)
__all__.extend(transforms.__all__)
