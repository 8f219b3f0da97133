"""Which call sites still cast f32 streams to operands / run channel_stats / small torch ops in one iteration (tuning)."""
import collections, os, sys, traceback, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["L2I_OVERLAP"] = "0"
import layout2img_amd as L
from layout2img_amd import ops
from layout2img_amd.synthetic import make_batch
dev = torch.device("cuda:0")
torch.manual_seed(1234)
netG = L.ResnetGenerator128_context(num_classes=184).finalize(dev, torch.bfloat16)
netD = L.CombineDiscriminator128_app(num_classes=184).finalize(dev, torch.bfloat16)
tr = L.GanTrainer(netG, netD)
batch = make_batch(32, 128, "coco", seed=1234, device=dev)[:4]
tr.step(*batch)
cnt = collections.Counter()
def wrap(name):
    orig = getattr(ops, name)
    def f(*a, **k):
        st = traceback.extract_stack()[:-1]
        site = " < ".join(f"{os.path.basename(s.filename)}:{s.lineno}:{s.name}" for s in st[-4:-1][::-1])
        x = a[0]
        cnt[(name, tuple(x.shape), site)] += 1
        return orig(*a, **k)
    setattr(ops, name, f)
for n in ("cast_op", "channel_stats"):
    wrap(n)
tr.step(*batch)
torch.cuda.synchronize()
for (name, shape, site), n in sorted(cnt.items(), key=lambda kv: -kv[1] * torch.Size(kv[0][1]).numel()):
    print(f"{n:3d} x {name:14s} {str(shape):28s} {site}")
