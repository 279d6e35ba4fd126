"""Import shim: the package directory is ``mpi-bicgstab_amd/`` (not a valid Python identifier);
this makes its ``python/`` sub-directory importable as ``mpi_bicgstab_amd``."""
import os as _os

__path__.append(_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                              "mpi-bicgstab_amd", "python"))
