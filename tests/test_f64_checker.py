"""The float64 checker of the benchmark network (tools/f64_check.py) writes its backward pass out by hand (it runs at
125 M ratings, where autograd's saved tensors would not fit).  Here it is pinned on a small graph against torch
autograd over the dense per-level adjacency matrices -- forward values, loss and every gradient."""
import numpy as np
import torch

from tools import f64_check as F


def _case(seed=0, nu=23, ni=11, R=3, D=6, U=5, O=4, P=3, E=90):
    rng = np.random.default_rng(seed)
    keys = rng.choice(nu * ni, size=E, replace=False)
    keys.sort()
    u, i = keys // ni, keys % ni
    ind_ptr = np.zeros(nu + 1, np.int64)
    np.cumsum(np.bincount(u, minlength=nu), out=ind_ptr[1:])
    level = rng.integers(0, R, E)
    g = torch.Generator().manual_seed(seed)
    mk = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64) * 0.5
    widths = [D, O]
    layers = []
    for l in range(2):
        lp = {}
        for k in ("user", "item"):
            lp[k] = {"W": [mk(U, widths[l]) for _ in range(R)], "b": [mk(U) for _ in range(R)], "Wo": mk(O, U), "bo": mk(O)}
        layers.append(lp)
    params = {"embed": {"user": mk(nu, D), "item": mk(ni, D)}, "layers": layers,
              "proj": {"user": (mk(P, O), mk(P)), "item": (mk(P, O), mk(P))}}
    y = mk(E)
    return (torch.from_numpy(ind_ptr), torch.from_numpy(i), torch.from_numpy(level), ni, R), u, i, level, params, y


def _leaves(params):
    out = [params["embed"]["user"], params["embed"]["item"]]
    for lp in params["layers"]:
        for k in ("user", "item"):
            out += lp[k]["W"] + lp[k]["b"] + [lp[k]["Wo"], lp[k]["bo"]]
    for k in ("user", "item"):
        out += list(params["proj"][k])
    return out


def test_checker_matches_dense_autograd():
    arrays, u, i, level, params, y = _case()
    nu, ni, R = arrays[0].numel() - 1, arrays[3], arrays[4]
    g = F.RawGraph(*arrays, chunk=17)                          # several chunks
    scale = 1.0 / y.numel()
    res = F.evaluate(g, params, y, scale)
    # dense definition with autograd
    du, di = np.bincount(u, minlength=nu), np.bincount(i, minlength=ni)
    A = torch.zeros(R, nu, ni, dtype=torch.float64)
    w = np.sqrt(np.float32(1.0) / du[u].astype(np.float32) / di[i].astype(np.float32)).astype(np.float64)
    A[torch.from_numpy(level), torch.from_numpy(u), torch.from_numpy(i)] = torch.from_numpy(w)
    # the item->user matrix divides in the other order (fp32): may differ from w in the last bit
    wt = np.sqrt(np.float32(1.0) / di[i].astype(np.float32) / du[u].astype(np.float32)).astype(np.float64)
    At = torch.zeros(R, ni, nu, dtype=torch.float64)
    At[torch.from_numpy(level), torch.from_numpy(i), torch.from_numpy(u)] = torch.from_numpy(wt)
    for t in _leaves(params):
        t.requires_grad_(True)
    leaky = lambda v: torch.where(v > 0, v, 0.1 * v)
    x = {"user": params["embed"]["user"], "item": params["embed"]["item"]}
    outs = []
    for lp in params["layers"]:
        hu = leaky(sum(A[r] @ (x["item"] @ lp["user"]["W"][r].t() + lp["user"]["b"][r]) for r in range(R)))
        hi = leaky(sum(At[r] @ (x["user"] @ lp["item"]["W"][r].t() + lp["item"]["b"][r]) for r in range(R)))
        x = {"user": leaky(hu @ lp["user"]["Wo"].t() + lp["user"]["bo"]),
             "item": leaky(hi @ lp["item"]["Wo"].t() + lp["item"]["bo"])}
        outs.append(x)
    pu = x["user"] @ params["proj"]["user"][0].t() + params["proj"]["user"][1]
    pi = x["item"] @ params["proj"]["item"][0].t() + params["proj"]["item"][1]
    score = (pu[torch.from_numpy(u)] * pi[torch.from_numpy(i)]).sum(1)
    loss = scale * (0.5 * (score - y) ** 2).sum()
    loss.backward()
    close = lambda a, b: torch.allclose(a, b.detach(), rtol=1e-11, atol=1e-13)
    assert close(res["loss"], loss)
    assert abs(res["score_rms"] - float(score.detach().pow(2).mean().sqrt())) < 1e-12
    for l in range(2):
        for k in ("user", "item"):
            assert close(res["layer_out"][l][k], outs[l][k])
            gl, lp = res["grads"]["layers"][l][k], params["layers"][l][k]
            for r in range(R):
                assert close(gl["W"][r], lp["W"][r].grad) and close(gl["b"][r], lp["b"][r].grad)
            assert close(gl["Wo"], lp["Wo"].grad) and close(gl["bo"], lp["bo"].grad)
    for k in ("user", "item"):
        assert close(res["proj"][k], {"user": pu, "item": pi}[k])
        assert close(res["grads"]["proj"][k][0], params["proj"][k][0].grad)
        assert close(res["grads"]["proj"][k][1], params["proj"][k][1].grad)
        assert close(res["grads"]["embed"][k], params["embed"][k].grad)


def test_checker_support_uses_override_item_degrees():
    arrays, u, i, level, params, y = _case(seed=3)
    ni = arrays[3]
    deg = torch.from_numpy(np.bincount(i, minlength=ni) + 2)
    g = F.RawGraph(*arrays, item_degrees=deg)
    du = np.bincount(u)
    want = np.sqrt(np.float32(1.0) / du[u].astype(np.float32) / deg.numpy()[i].astype(np.float32))
    assert np.array_equal(g.w_ui.numpy(), want.astype(np.float32))
