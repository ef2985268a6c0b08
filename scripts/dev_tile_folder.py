"""Developer probe: a folder's worth of 256 x 256 tiles through the tile manager, one by one vs sharing batches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cerberus_amd.tile import InferManager
from cerberus_amd.weights import DEFAULT_REQ_TARGET_CODE, default_model_kwargs
mgr = InferManager(checkpoint_path=None, decoder_dict=dict(DEFAULT_REQ_TARGET_CODE), model_args=default_model_kwargs())
rs = np.random.RandomState(0)
imgs = [rs.randint(0, 256, (256, 256, 3)).astype(np.uint8) for _ in range(128)]
mgr.infer_images(imgs[:32], 256, 256, 32); torch.cuda.synchronize()
t0 = time.time()
for im in imgs: mgr.infer_image(im, 256, 256, 32)
torch.cuda.synchronize(); t1 = time.time() - t0
t0 = time.time()
for i in range(0, 128, 32): mgr.infer_images(imgs[i:i + 32], 256, 256, 32)
torch.cuda.synchronize(); t2 = time.time() - t0
px = 128 * 256 * 256
print("128 tiles of 256^2 incl. post-processing: one by one %.3f s (%.1f Mpx/s), sharing batches of 32 %.3f s (%.1f Mpx/s)" % (t1, px / t1 / 1e6, t2, px / t2 / 1e6))
