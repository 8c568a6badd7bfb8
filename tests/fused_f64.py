"""Size-independent checker for fused_deep_moe at sizes the NumPy oracle cannot reach (BASELINE C5: 4096 tokens per rank, hidden
7168, 2I = 4096, 32 local experts per rank): a from-scratch float64 evaluation of the network for a SAMPLE of this rank's tokens,

    y[t] = sum_k w[t, k] * FFN_{e(t, k)}(x[t]),   FFN_e(x) = W2_e . requant(SwiGLU(W13_e . quant(x)))

with the kernel's rounding points mirrored (INT8 quantisation of x and of the SwiGLU output in fp32, bf16 expert output, fp32
k-ascending weighted sum -> bf16) -- the per-token formulation tests/test_oracle_ep.py pins the staged oracle against
(test_fused_deep_moe_oracle_against_independent_float64_pipeline), with the integer products taken exactly through torch float64
matmuls so that 256 tokens x 8 experts of DeepSeek-V3 size take seconds.  No dispatch tables, no HIP kernel of this repo, no int8 arithmetic: routing, ordering
and indexing mistakes of the product path cannot cancel against it.  Bars = the reference test's own
(tests/python/deepep/test_fused_deep_moe.py:470: avg relative diff < 4e-4; calc_diff as utils.py:191-215).
Test infrastructure only (imported by tests/ and by bench.py's validation of its C5 section)."""
import torch


def fused_weights(seed, L, H, I, device="cuda"):
    """Per-rank expert weights as tests/python/deepep/test_fused_deep_moe.py:32-44 draws them (randint(-16, 16), scales
    U * 4e-4 + 1.5e-3), in ORIGINAL column order (rows [0, I) of w13 = gate, [I, 2I) = up).  -> w13 [L, 2I, H] int8, w2 [L, H, I]
    int8, s13 [L, 2I] f32, s2 [L, H] f32."""
    g = torch.Generator(device=device).manual_seed(seed)
    w13 = torch.randint(-16, 16, (L, 2 * I, H), generator=g, device=device, dtype=torch.int32).to(torch.int8)
    w2 = torch.randint(-16, 16, (L, H, I), generator=g, device=device, dtype=torch.int32).to(torch.int8)
    s13 = torch.rand((L, 2 * I), generator=g, device=device) * 4e-4 + 1.5e-3
    s2 = torch.rand((L, H), generator=g, device=device) * 4e-4 + 1.5e-3
    return w13, w2, s13, s2


def fusion_perm(n, tile=128, device="cuda"):
    """permuted column j*tile + h*(tile/2) + i  <-  original column h*(n/2) + j*(tile/2) + i (oracle.ep.permute_fusion_cols)."""
    half = tile // 2
    j = torch.arange(n // tile, device=device)[:, None, None]
    h = torch.arange(2, device=device)[None, :, None]
    i = torch.arange(half, device=device)[None, None, :]
    return (h * (n // 2) + j * half + i).reshape(-1)


def _div32(a, b):
    """Correctly rounded fp32 quotient.  torch's fp32 division on the GPU is NOT (measured on MI355X: 127.0f / amax one ulp off in
    ~10 % of the rows, which flips first-quantisation steps of bf16 inputs); the quotient is formed in float64 and rounded once."""
    a = a.double() if torch.is_tensor(a) else a
    b = b.double() if torch.is_tensor(b) else b
    return (a / b).float()


@torch.no_grad()
def sampled_reference(x, idx, w, weights_of, L, sel):
    """Per-token evaluation (exact integer products, fp32 epilogues) of the tokens `sel` (1-D long tensor) of this rank.  weights_of(r) -> (w13, w2, s13, s2) of rank r
    (original column order).  -> [len(sel), H] float64."""
    xs = x[sel].float()
    S, H = xs.shape
    E_sel = idx[sel].long()
    # first quantisation point, fp32 as the kernel has it (moe_distribute_dispatch_v2.h:1006-1033): s = 127 / max|x|, q = rint(x * s)
    amax = xs.abs().amax(dim=1, keepdim=True)
    s = torch.where(amax > 0, _div32(127.0, amax), torch.zeros_like(amax))
    q = torch.round(xs * s).double()                                                       # torch.round: half to even
    sc = torch.where(amax > 0, _div32(1.0, s), torch.zeros_like(amax))                     # fp32 token scale
    Y = torch.zeros((S, E_sel.shape[1], H), dtype=torch.float32, device=x.device)      # bf16-rounded expert outputs per selection
    owners = torch.div(E_sel, L, rounding_mode="floor")
    for r in sorted(set(owners[E_sel >= 0].tolist())):
        w13, w2, s13, s2 = weights_of(int(r))
        I = w2.shape[2]
        for le in range(L):
            rows, ks = torch.nonzero(E_sel == r * L + le, as_tuple=True)
            if rows.numel() == 0:
                continue
            # integer products exactly (float64 matmul of integers < 2^53), then the fp32 epilogues with the kernel's rounding points:
            # d = (float(c) * w_scale[col]) * tok_scale[row]; v = up * gate / (1 + exp(-gate)); q2 = rint((v * 127) * (1 / rowmax))
            c = (q[rows] @ w13[le].double().t()).float()
            d = (c * s13[le].float()[None, :]) * sc[rows]
            gate, up = d[:, :I], d[:, I:]
            v = up * _div32(gate, 1 + torch.exp(-gate))
            vmax = v.abs().amax(dim=1, keepdim=True)
            inv = torch.where(vmax > 0, _div32(1.0, vmax), torch.zeros_like(vmax))
            q2 = torch.round((v * 127.0) * inv).double()
            c2 = (q2 @ w2[le].double().t()).float()
            y = (c2 * s2[le].float()[None, :]) * _div32(vmax, 127.0)
            Y[rows, ks] = y.to(torch.bfloat16).float()       # an expert's output row leaves its GEMM2 as bf16
        del w13, w2, s13, s2
    # the weighted sum as the combine kernel forms it (cam_moe_combine_normal.h:359-400): fp32, separate multiply and add, k ascending
    acc = torch.zeros((S, H), dtype=torch.float32, device=x.device)
    w32 = w[sel].float()
    for k in range(E_sel.shape[1]):
        valid = (E_sel[:, k] >= 0)[:, None]
        acc = torch.where(valid, acc + Y[:, k] * w32[:, k:k + 1], acc)
    return acc.to(torch.bfloat16).double()


def diffs(got, want):
    """-> (calc_diff, avg relative diff) with the reference test's definitions."""
    a, b = got.double(), want.double()
    denom = (a * a + b * b).sum()
    calc = float(1 - 2 * (a * b).sum() / denom) if float(denom) > 0 else 0.0
    avg = float(((a - b).abs() / b.abs().clamp_min(1e-2)).mean())
    return calc, avg


def sampled_check(out, x, idx, w, weights_of, L, n_samples=256, seed=0):
    """-> dict(calc_diff, avg_diff, samples, ok)."""
    T = x.shape[0]
    g = torch.Generator(device="cpu").manual_seed(seed)
    sel = torch.randperm(T, generator=g)[:min(n_samples, T)].to(x.device)
    want = sampled_reference(x, idx, w, weights_of, L, sel)
    calc, avg = diffs(out[sel], want)
    return {"calc_diff": calc, "avg_diff": avg, "samples": int(sel.numel()), "ok": bool(calc < 1e-5 and avg < 4e-4)}
