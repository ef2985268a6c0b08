"""cerberus_amd -- MI355X (gfx950) native tiled-inference hot path of Cerberus behind the reference's own API.

Host mirrors of the reference interface (same names / argument meaning):
  cerberus_amd.net_desc.create_model / NetDesc      <- reference models/net_desc.py
  cerberus_amd.run_desc.infer_step                  <- reference models/run_desc.py:439-502
  cerberus_amd.postproc.PostProcInstErodedContourMap <- reference loader/postproc.py:268-407
  cerberus_amd.tile / cerberus_amd.wsi              <- reference infer/tile.py, infer/wsi.py (geometry + stitching)
All arithmetic runs in libcerberus_hip.so (include/cerberus_hip.h); there is no CPU fallback.
"""
__version__ = "0.1.0"
