"""MI355X-native sensor-update + resample engine behind badger_amcl's pf / laser-sensor API.

The compute lives in libbadger_pf_hip.so (hand-written HIP for gfx950, C-ABI in
include/badger_pf.h).  Importing this package does not load the library; constructing an
`Engine` does, and raises if it is missing -- there is no CPU fallback.
"""
from .pf import (BpfError, Engine, OccupancyMap, OctoMap, Odom, OdomData, ParticleFilter, PFSampleSet,  # noqa: F401
                 PlanarData, PlanarScanner, PointCloudData, PointCloudScanner)
