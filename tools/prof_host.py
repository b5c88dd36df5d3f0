import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import squeezedet_amd as S
from squeezedet_amd import nets, synthetic, ops
from squeezedet_amd.train import SqueezeDetTrainer
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_train import synthetic_ground_truth
mc = S.kitti_squeezeDet_config(); mc.LOAD_PRETRAINED_MODEL=False; mc.IS_TRAINING=True; mc.BATCH_SIZE=20
m = nets.SqueezeDet(mc, gpu_id="0", dtype=torch.float16); m.load_params(synthetic.synthetic_params(m, seed=0))
tr = SqueezeDetTrainer(m, lazy_overflow_check=True)
dev = "cuda:0"
x = synthetic.synthetic_images(20, mc.IMAGE_HEIGHT, mc.IMAGE_WIDTH, seed=100).to(dev)
anchors = torch.from_numpy(np.asarray(mc.ANCHOR_BOX, np.float64)).to(dev)
gt, gcls, gcnt = [torch.from_numpy(a).to(dev) for a in synthetic_ground_truth(mc, 20, seed=200)]
nobj = float(gcnt.sum().item())
step = lambda: tr.step(x, *ops.build_labels(anchors, gt, gcls, gcnt, mc.CLASSES)[:4], num_objects=nobj)
for _ in range(5): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:4500])
