"""Generates tests/golden/ep_case_*.npz: small seeded dispatch/combine cases (inputs + every output, including the
intermediate index tables) from the CPU oracle, as planned in SURVEY.md section 8(c) "Fixtures we commit".

The reference cannot run here (AscendC + CANN), so these vectors do not come from a reference execution: they freeze the
oracle -- which tests/test_oracle_ep.py pins to the reference tests' closed-form goldens -- so that any later change of
the oracle or of the kernels shows up as a diff against committed data.

    python tests/golden/gen_ep_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ep as O                      # noqa: E402
from oracle.bf16 import f32_to_bf16_bits_rne    # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
#        name  W  T   H    K  E   drop  quant
CASES = [("a", 1, 33, 128, 2, 8, 0.0, True),
         ("b", 2, 17, 256, 4, 16, 0.2, True),
         ("c", 4, 9, 128, 8, 32, 0.1, False),
         ("d", 8, 4, 128, 1, 8, 0.0, True)]


def make(rng, W, T, H, K, E, drop):
    xs, idxs, ws = [], [], []
    for r in range(W):
        t = T + r                                               # ragged token counts
        xs.append(f32_to_bf16_bits_rne((rng.standard_normal((t, H)) * 2).astype(np.float32)))
        idx = np.argsort(-rng.random((t, E)), axis=1)[:, :K].astype(np.int64)
        idx[rng.random((t, K)) < drop] = -1
        idxs.append(idx)
        ws.append(rng.standard_normal((t, K)).astype(np.float32))
    return xs, idxs, ws


def main():
    for name, W, T, H, K, E, drop, quant in CASES:
        rng = np.random.default_rng({"a": 11, "b": 22, "c": 33, "d": 44}[name])
        xs, idxs, ws = make(rng, W, T, H, K, E, drop)
        disp = O.normal_dispatch(xs, idxs, E, quant)
        ys = [O.per_token_cast_back(d.recv_x, d.recv_x_scales) if quant else d.recv_x for d in disp]
        comb = O.combine(ys, [d.recv_src_idx for d in disp], [d.total_recv for d in disp], idxs, ws, E)
        ll = O.low_latency_dispatch(xs, idxs, T + W, E, quant)
        data = dict(W=np.int32(W), H=np.int32(H), K=np.int32(K), E=np.int32(E), quant=np.int32(quant), max_tokens=np.int32(T + W))
        for r in range(W):
            lay = O.dispatch_layout(idxs[r], E, W)
            data.update({f"x{r}": xs[r], f"idx{r}": idxs[r], f"w{r}": ws[r],
                         f"lay_num_tokens_per_rank{r}": lay["num_tokens_per_rank"],
                         f"lay_num_tokens_per_expert{r}": lay["num_tokens_per_expert"],
                         f"lay_is_token_in_rank{r}": lay["is_token_in_rank"],
                         f"lay_send_token_idx_small{r}": lay["send_token_idx_small"],
                         f"recv_x{r}": disp[r].recv_x[:max(disp[r].total_recv, 1)],
                         f"recv_scales{r}": (disp[r].recv_x_scales if quant else np.zeros(1, np.float32))[:max(disp[r].total_recv, 1)],
                         f"recv_src_idx{r}": disp[r].recv_src_idx[:3 * disp[r].total_recv],
                         f"send_head{r}": disp[r].send_head, f"per_expert_list{r}": np.asarray(disp[r].num_recv_tokens_per_expert_list),
                         f"combined{r}": comb[r],
                         f"ll_recv_count{r}": ll[r].packed_recv_count, f"ll_layout_range{r}": ll[r].layout_range,
                         f"ll_src_info{r}": ll[r].src_info, f"ll_recv_x{r}": ll[r].packed_recv_x[:max(ll[r].total, 1)]})
        np.savez_compressed(os.path.join(OUT, f"ep_case_{name}.npz"), **data)
        print(name, "ok", [d.total_recv for d in disp])


if __name__ == "__main__":
    main()
