"""The config-3 composition test's flow with a 2-step sampler (cheap oracle): where do sweep 1 and sweep 2 differ?"""
import gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from diff_unet_amos_amd import inference, engine
from test_diffunet_gpu import _pair, _seeded_predictor
from oracle.sliding_window_ref import sliding_window_ref

STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
STEPS_ = STEPS
mode = sys.argv[1] if len(sys.argv) > 1 else "base"
if mode == "nogc":
    gc.disable()
if mode in ("pre", "post", "mid"):
    import torch as _t
    orig_replay = _t.cuda.CUDAGraph.replay
    state = {"n": 0}

    def replay(self):
        if mode == "pre" and state["n"] % STEPS_ == 0:
            _t.cuda.synchronize()
        orig_replay(self)
        state["n"] += 1
        if mode == "post" and state["n"] % STEPS_ == 0:
            _t.cuda.synchronize()
        if mode == "mid" and state["n"] % STEPS_ == 1:
            _t.cuda.synchronize()
    _t.cuda.CUDAGraph.replay = replay
if mode == "eager":
    orig = engine.Plan.sample_loop
    engine.Plan.sample_loop = lambda self, *a, **k: orig(self, *a, **{**k, "use_graph": False})
dtype = torch.float32
kw = dict(in_channels=1, out_channels=16, features=(8, 8, 16, 32, 64, 8))
net, ref = _pair(kw, dtype, sample_steps=STEPS)
g = torch.Generator().manual_seed(31)
vol = torch.rand(1, 1, 48, 48, 40, generator=g)
shape = (1, 16, 32, 32, 32)


def ref_fn(win):
    w = torch.from_numpy(win).float()
    seed = int(w.cuda().double().abs().sum().item() * 1e3) % (2 ** 31)
    torch.manual_seed(seed)
    xT = torch.randn(*shape, device="cuda").cpu()
    with torch.no_grad():
        return ref.ddim_sample(w, x_T=[xT], step_noise=[[torch.zeros(shape)] * STEPS]).numpy()


want = torch.from_numpy(sliding_window_ref(vol.numpy(), (32, 32, 32), 0.25, ref_fn)).float()
wins = [[], [], []]


def pred(k):
    base = _seeded_predictor(net)

    def f(x, **kw2):
        o = base(x, **kw2)
        if mode == "sync":
            torch.cuda.synchronize()
        wins[k].append(o)          # keep a reference only: no extra kernel, no sync
        return o
    return f


with torch.no_grad():
    for k in range(3):
        inference.sliding_window_inference(vol.cuda(), (32, 32, 32), 1, pred(k), 0.25, pred_type="ddim_sample").cpu()
print(mode, "gc counts", gc.get_count(), "sweep 1/2 vs 0 per-window max |d|:",
      [[f"{float((a - b).abs().max()):.1e}" for a, b in zip(wins[0], wins[k])] for k in (1, 2)])
