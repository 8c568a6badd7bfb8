"""Generates tests/golden/mla_ref_fp16_*.npz by running the REFERENCE Triton kernel `decode_mla`
(/root/reference/python/sgl_kernel_npu/sgl_kernel_npu/attention/decode_attention.py) on CPU under
TRITON_INTERPRET=1 in the build container.  Only the resulting vectors (inputs + outputs) are committed; the
reference source never travels.  fp16 only: the Triton interpreter mishandles bf16.

    PYTHONDONTWRITEBYTECODE=1 TRITON_INTERPRET=1 python tests/golden/gen_mla_golden.py
"""
import importlib.util
import os
import sys

os.environ["TRITON_INTERPRET"] = "1"
sys.dont_write_bytecode = True
import numpy as np
import torch

REF = "/root/reference/python/sgl_kernel_npu/sgl_kernel_npu/attention/decode_attention.py"
OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    spec = importlib.util.spec_from_file_location("ref_decode_attention", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    # a-c: kv groups of <= 64 heads (mla_decode_kernel); d-f: groups of 128 heads -- the kernel BASELINE C4 runs
    # (mla_decode_wide_kernel): C4's own head layout (128 q heads on one latent head), two kv heads of 128, a short ragged batch
    cases = [("a", 2, 16, 1, 200, 64, 7), ("b", 3, 32, 1, 131, 16, 11), ("c", 1, 8, 1, 64, 64, 3),
             ("d", 2, 128, 1, 300, 64, 13), ("e", 1, 256, 2, 130, 64, 17), ("f", 3, 128, 1, 77, 16, 19)]
    for name, B, Hq, Hkv, S, page, seed in cases:
        if os.path.exists(os.path.join(OUT, f"mla_ref_fp16_{name}.npz")) and "--all" not in sys.argv:
            continue
        torch.manual_seed(seed)
        max_pages = (S + page - 1) // page
        nblocks = B * max_pages + 2
        q = torch.randn((B, Hq, 576), dtype=torch.float16)
        k_nope = torch.randn((nblocks, page, Hkv, 512), dtype=torch.float16)
        k_rope = torch.randn((nblocks, page, Hkv, 64), dtype=torch.float16)
        perm = torch.randperm(nblocks)[:B * max_pages].to(torch.int32).reshape(B, max_pages)
        lens = torch.tensor([S - 17 * i for i in range(B)], dtype=torch.int32).clamp(min=1)
        out = torch.zeros((B, Hq, 512), dtype=torch.float16)
        sm_scale = 1.0 / (576 ** 0.5)
        mod.decode_mla(q, k_nope, k_rope, out, lens, sm_scale, page, perm)
        np.savez_compressed(os.path.join(OUT, f"mla_ref_fp16_{name}.npz"), q=q.numpy(), k_nope=k_nope.numpy(),
                            k_rope=k_rope.numpy(), block_table=perm.numpy(), kv_seq_lens=lens.numpy(),
                            sm_scale=np.float32(sm_scale), page_size=np.int32(page), out=out.numpy())
        print(name, "ok", float(out.abs().max()))


if __name__ == "__main__":
    main()
