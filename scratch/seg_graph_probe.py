"""Probe (round 4): can a backward pass be captured as several HIP graphs, cut from inside a post-accumulate-grad hook?

The hook runs on autograd's device thread, so the cut (capture_end of graph k, capture_begin of graph k+1 on the same
stream, same memory pool) happens on a different thread than the first capture_begin: needs capture_error_mode="relaxed".
Prints what works; compares the segmented replay with the eager step."""
import threading

import torch

dev = torch.device("cuda:0")
torch.manual_seed(0)
layers = [torch.nn.Linear(256, 256).to(dev) for _ in range(6)]
params = [p for l in layers for p in l.parameters()]
x = torch.randn(64, 256, device=dev)


def fwd(x):
    h = x
    for l in layers:
        h = torch.relu(l(h))
    return (h * h).mean()


def eager():
    for p in params:
        p.grad = None
    loss = fwd(x)
    loss.backward()
    return loss.detach().clone(), [p.grad.clone() for p in params]


loss0, g0 = eager()
torch.cuda.synchronize()
print("main thread", threading.get_ident())

cut_after = {id(layers[4].weight), id(layers[2].weight)}      # two cuts -> three backward segments
graphs = []
state = {"cur": None, "pool": None, "stream": None, "threads": set(), "err": None}


def hook(p):
    state["threads"].add(threading.get_ident())
    if id(p) in cut_after and state["err"] is None:
        try:
            cur = torch.cuda.current_stream()
            assert cur == state["stream"], (cur, state["stream"])
            state["cur"].capture_end()
            g = torch.cuda.CUDAGraph()
            g.capture_begin(pool=state["pool"], capture_error_mode="relaxed")
            graphs.append(g)
            state["cur"] = g
        except Exception as e:          # noqa: BLE001
            state["err"] = repr(e)


handles = [p.register_post_accumulate_grad_hook(hook) for p in params]
for p in params:
    p.grad = None
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
torch.cuda.synchronize()
with torch.cuda.stream(s):
    g = torch.cuda.CUDAGraph()
    state["stream"] = s
    state["pool"] = torch.cuda.graph_pool_handle()
    g.capture_begin(pool=state["pool"], capture_error_mode="relaxed")
    state["cur"] = g
    graphs.append(g)
    loss = fwd(x)
    loss.backward()
    out_loss = loss.detach()
    state["cur"].capture_end()
torch.cuda.current_stream().wait_stream(s)
print("hook threads", state["threads"], "err", state["err"], "graphs", len(graphs))
for h in handles:
    h.remove()
grads = [p.grad for p in params]
for rep in range(3):
    for gr in grads:
        gr.zero_()
    for g in graphs:
        g.replay()
    torch.cuda.synchronize()
    ok = all(torch.equal(a, b) for a, b in zip(grads, g0)) and torch.equal(out_loss, loss0)
    print("replay", rep, "equal to eager:", ok, float(out_loss), float(loss0))

# partial replay: only the first graph -> only the gradients of layers 5 and 4 (weight) are written
for gr in grads:
    gr.zero_()
graphs[0].replay()
torch.cuda.synchronize()
print("after segment 0 only: nonzero grads", [bool(gr.abs().sum() > 0) for gr in grads])
