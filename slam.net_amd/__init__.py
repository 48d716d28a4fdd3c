"""slam.net_amd -- MI355X-native CoreSLAM / HectorSLAM hot path (drop-in for mikkleini/slam.net).

The product is the C-ABI shared library ``libslamhip.so`` (include/slamhip.h) built from
``csrc/`` with hipcc for gfx950; this package is the thin host side used by tests and benches:

  capi      ctypes binding of every symbol in include/slamhip.h (loads the .so, fails loudly)
  coreslam  host mirror of CoreSLAM.CoreSLAMProcessor / HoleMap / ObstacleMap
  hector    host mirror of HectorSLAM ScanMatcher / OccGridMap / MapRepMultiMap / HectorSLAMProcessor
  sim       synthetic world + lidar scan generator (inputs only)
  build     hipcc build recipe for libslamhip.so

There is NO CPU fallback: every compute entry point goes through the HIP library.
"""
__version__ = "0.1.0"
