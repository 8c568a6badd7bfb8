"""TEST DOUBLE for deep_ep_cpp.Buffer on CPU tensors, backed by the oracle.

Used ONLY by the CPU/gloo plumbing tests to exercise the host logic of the `alltoall` strategies and of
deep_ep.Buffer (argument normalisation, handle packing, split computation, collectives) where no GPU exists.
The product never imports this; on a GPU box the real deep_ep_cpp runtime runs the same strategy code."""
import numpy as np
import torch

from oracle import ep as O
from oracle.bf16 import bits_to_torch, torch_to_bits

META = 16


class FakeRuntime:
    def __init__(self, rank, num_ranks, *_):
        self.rank, self.W = rank, num_ranks
        self._lay = None

    def is_available(self):
        return True

    def get_num_rdma_ranks(self):
        return 1

    # ---- layout
    def _layout(self, topk_idx, E):
        l = O.dispatch_layout(topk_idx.numpy().astype(np.int64), E, self.W)
        l["send_data_offset"] = O.send_data_offset(l["num_tokens_per_expert"])
        return l

    def get_dispatch_layout(self, topk_idx, num_experts, *_):
        l = self._lay = self._layout(topk_idx, num_experts)
        t = torch.from_numpy
        return t(l["num_tokens_per_rank"]), None, t(l["num_tokens_per_expert"]), t(l["is_token_in_rank"]), None

    # ---- a2a entry points (same contracts as deep_ep_cpp.Buffer.a2a_*)
    def a2a_dispatch_stage(self, x, topk_idx, num_experts, quant_type):
        E = num_experts
        l = self._layout(topk_idx, E)
        T, K = topk_idx.shape
        H = x.shape[1]
        xb = torch_to_bits(x)
        quant = quant_type != "bf16"
        pay = H if quant else 2 * H
        rows = np.zeros((max(T * K, 1), pay + META), np.uint8)
        if quant:
            q, s = O.quant_int8_rows(xb, None if quant_type == "int8_ll" else 1e-12)
        ti = topk_idx.numpy().astype(np.int64)
        for t in range(T):
            for k in range(K):
                e = ti[t, k]
                if e < 0 or e >= E:
                    continue
                slot = l["send_data_offset"][e] + l["send_token_idx_small"][t, k]
                if quant:
                    rows[slot, :H] = q[t].view(np.uint8)
                    rows[slot, pay:pay + 4] = np.array([s[t]], np.float32).view(np.uint8)
                else:
                    rows[slot, :pay] = xb[t].view(np.uint8)
                rows[slot, pay + 4:pay + 16] = np.array([t, k, self.rank], np.int32).view(np.uint8)
        cnt = np.concatenate([l["num_tokens_per_expert"], [T]]).astype(np.int32)
        return torch.from_numpy(rows), torch.from_numpy(cnt)

    def a2a_dispatch_tables(self, cnt_matrix):
        c = cnt_matrix.numpy().astype(np.int64)
        W, E = self.W, c.shape[1] - 1
        L = E // W
        nt = O.notify_dispatch(c[:, :E], c[:, E].tolist(), self.rank)
        send_off_all = np.zeros((W, E), np.int64)
        np.cumsum(c[:, :E - 1], axis=1, out=send_off_all[:, 1:])
        pull = nt["recv_offset"].reshape(L, W) - send_off_all[:, self.rank * L][None, :]
        send_rows = [int(c[self.rank, r * L:(r + 1) * L].sum()) for r in range(W)]
        recv_rows = [int(c[r, self.rank * L:(self.rank + 1) * L].sum()) for r in range(W)]
        return (torch.from_numpy(nt["recv_count"]), torch.from_numpy(pull.reshape(-1).astype(np.int32)), send_rows, recv_rows,
                [int(v) for v in nt["recv_tokens_per_expert"]], int(nt["total_recv_token"]), int(nt["max_bs"]))

    def a2a_dispatch_unpack(self, staging, recv_rows, recv_count, pull_offset, hidden, total_recv, quant_type, min_rows,
                            src_idx_len):
        W = self.W
        st = staging.numpy()
        H = hidden
        quant = quant_type != "bf16"
        pay = H if quant else 2 * H
        rows = max(max(total_recv, 1), min_rows)
        rx = np.zeros((rows, pay), np.uint8)
        rs = np.zeros(rows, np.float32)
        tri = np.zeros(max(rows * 3, src_idx_len), np.int32)
        base = np.concatenate([[0], np.cumsum(recv_rows)])
        cum = recv_count.numpy()
        po = pull_offset.numpy()
        prev = 0
        for i in range(cum.size):
            src = i % W
            for j in range(cum[i] - prev):
                r = prev + j
                row = st[base[src] + po[i] + j]
                rx[r] = row[:pay]
                rs[r] = row[pay:pay + 4].view(np.float32)[0]
                t, k, s = row[pay + 4:pay + 16].view(np.int32)
                tri[3 * r:3 * r + 3] = (s, t, k)
            prev = cum[i]
        if quant:
            return torch.from_numpy(rx.view(np.int8)), torch.from_numpy(rs), torch.from_numpy(tri)
        return bits_to_torch(rx.view(np.uint16)), None, torch.from_numpy(tri)

    def a2a_combine_pack(self, x, send_head):
        W = self.W
        cum = send_head.numpy()
        L = cum.size // W
        c = np.diff(np.concatenate([[0], cum])).reshape(L, W)
        xb = torch_to_bits(x)
        out = np.zeros_like(xb)
        rows_per_src = c.sum(axis=0)
        blk = np.concatenate([[0], np.cumsum(rows_per_src)])
        for src in range(W):
            run = blk[src]
            for le in range(L):
                i = le * W + src
                start = cum[i] - c[le, src]
                out[run:run + c[le, src]] = xb[start:start + c[le, src]]
                run += c[le, src]
        return bits_to_torch(out), [int(v) for v in rows_per_src]

    def a2a_combine_prepare(self, topk_idx, num_experts):
        l = self._layout(topk_idx, num_experts)
        L = num_experts // self.W
        rows = [int(l["num_tokens_per_expert"][r * L:(r + 1) * L].sum()) for r in range(self.W)]
        return torch.from_numpy(l["send_data_offset"]), torch.from_numpy(l["send_token_idx_small"]), rows

    def a2a_combine_reduce(self, returned, topk_idx, topk_weights, send_off, idx_small, hidden, num_experts):
        ti = topk_idx.numpy().astype(np.int64)
        T, K = ti.shape
        valid = (ti >= 0) & (ti < num_experts)
        slots = np.where(valid, send_off.numpy()[np.clip(ti, 0, num_experts - 1)] + idx_small.numpy(), 0)
        rb = torch_to_bits(returned)
        from oracle.bf16 import bf16_bits_to_f32
        rows = bf16_bits_to_f32(rb[slots.reshape(-1)]).reshape(T, K, hidden)
        w = np.ones((T, K), np.float32) if topk_weights is None else topk_weights.numpy()
        return bits_to_torch(O.weighted_reduce(rows, valid, w))
