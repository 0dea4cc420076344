"""Experimental routines (reference pulsarbat/contrib/__init__.py)."""

from .misc import *
from . import misc

__all__ = misc.__all__.copy()
