"""Readers (reference pulsarbat/readers/__init__.py): ``BaseReader`` plus baseband-file readers whose payload
unpacking runs on the GPU (``pbh_decode``), so that raw 2/8-bit samples cross PCIe instead of complex64
(SURVEY.md 8f rank 4).  File-format parsing is host work in ``_formats`` (the reference delegates it to the
third-party ``baseband`` package, which is absent here)."""

from ._base import BaseReader, OutOfBoundsError
from ._baseband_readers import BasebandReader, GUPPIRawReader, DADAStokesReader

__all__ = ["BaseReader", "OutOfBoundsError", "BasebandReader", "GUPPIRawReader", "DADAStokesReader"]
