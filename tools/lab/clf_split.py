"""The classifier's forward + backward pass (ResNet-18, 240 x 240 crop of 256 x 256) at batch 64 as ONE chain of launches against TWO
half-batch chains on two HIP streams.  Why: the per-layer table says a launch of the small-map layers costs ~30 us whatever the batch
(batch 32: 1140 us of convolution launches, batch 64: 1473) -- the launches are latency-bound, so two independent half-batch chains might
fill each other's gaps.

    python tools/lab/clf_split.py [f32|f16]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from spaa_amd import synthetic as syn  # noqa: E402
from spaa_amd.classifier import Classifier  # noqa: E402

storage = sys.argv[1] if len(sys.argv) > 1 else 'f32'
dev = torch.device('cuda:0')
B, H = 64, 256
csd = syn.resnet18_state_dict(2, logit_gain=20.0)
clf = Classifier('resnet18', dev, state_dict=csd)
owner = [object(), type('o', (), {})(), type('o', (), {})(), type('o', (), {})()]
full = clf.engine(B, (H, H), (240, 240), owner=owner[1], storage=storage)
half = [clf.engine(B // 2, (H, H), (240, 240), owner=owner[2], storage=storage), clf.engine(B // 2, (H, H), (240, 240), owner=owner[3], storage=storage)]
assert half[0] is not half[1]
y = torch.rand(B, H, H, 4, device=dev)
g = torch.randn(B, full.ncls, device=dev) * 1e-2
side = torch.cuda.Stream(device=dev)
ev_f, ev_j = torch.cuda.Event(), torch.cuda.Event()


def one():
    full.forward(y)
    full.backward(g)


def two_seq():
    for i in range(2):
        half[i].forward(y[i * 32:(i + 1) * 32])
        half[i].backward(g[i * 32:(i + 1) * 32])


def two_par():
    main = torch.cuda.current_stream(dev)
    ev_f.record(main)
    side.wait_event(ev_f)
    with torch.cuda.stream(side):
        half[1].forward(y[32:])
        half[1].backward(g[32:])
        ev_j.record(side)
    half[0].forward(y[:32])
    half[0].backward(g[:32])
    main.wait_event(ev_j)


def two_par_layerwise_fwd_then_bwd():
    # fork / join around the forward passes and again around the backward passes (what the attack loop needs: the decision sits between)
    main = torch.cuda.current_stream(dev)
    for fn in ('forward', 'backward'):
        ev_f.record(main)
        side.wait_event(ev_f)
        with torch.cuda.stream(side):
            getattr(half[1], fn)(y[32:] if fn == 'forward' else g[32:])
            ev_j.record(side)
        getattr(half[0], fn)(y[:32] if fn == 'forward' else g[:32])
        main.wait_event(ev_j)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for name, fn in (('one chain, batch 64', one), ('two half-batch chains, one stream', two_seq), ('two half-batch chains, two streams', two_par),
                 ('two streams, joined between forward and backward', two_par_layerwise_fwd_then_bwd)):
    print(f'{storage} {name:52s} {timeit(fn):8.1f} us', flush=True)
