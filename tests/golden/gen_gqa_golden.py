"""Generates tests/golden/gqa_ref_fp16_*.npz by running the REFERENCE Triton kernel `decode_gqa`
(/root/reference/python/sgl_kernel_npu/sgl_kernel_npu/attention/decode_attention.py:378-450) on CPU under
TRITON_INTERPRET=1 in the build container.  Only the resulting vectors (inputs + outputs) are committed; the
reference source never travels.  fp16 only (the Triton interpreter mishandles bf16).  The reference kernel takes
page_size as its key tile, so pages are powers of two here; head dims too (upstream Triton's arange rejects the
reference's 288 / 576 = 256+32 / 512+64 shapes that Triton-Ascend accepts).

    PYTHONDONTWRITEBYTECODE=1 TRITON_INTERPRET=1 python tests/golden/gen_gqa_golden.py
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
    #        name B  Hq Hkv Lk   Lv   S   page seed v_is_view
    cases = [("a", 2, 16, 2, 128, 128, 150, 32, 5, False),
             ("b", 2, 8, 1, 256, 128, 90, 16, 9, True),
             ("c", 1, 8, 8, 64, 64, 70, 64, 13, False)]
    for name, B, Hq, Hkv, Lk, Lv, S, page, seed, view in cases:
        torch.manual_seed(seed)
        max_pages = (S + page - 1) // page
        nblocks = B * max_pages + 2
        q = torch.randn((B, Hq, Lk), dtype=torch.float16)
        k = torch.randn((nblocks, page, Hkv, Lk), dtype=torch.float16)
        v = k[..., :Lv] if view else torch.randn((nblocks, page, Hkv, Lv), dtype=torch.float16)
        perm = torch.randperm(nblocks)[:B * max_pages].to(torch.int32).reshape(B, max_pages)
        lens = torch.tensor([S - 23 * i for i in range(B)], dtype=torch.int32).clamp(min=1)
        out = torch.zeros((B, Hq, Lv), dtype=torch.float16)
        sm_scale = 1.0 / (Lk ** 0.5)
        mod.decode_gqa(q, k, v, out, lens, sm_scale, page, perm)
        np.savez_compressed(os.path.join(OUT, f"gqa_ref_fp16_{name}.npz"), q=q.numpy(), k=k.numpy(),
                            v=np.ascontiguousarray(v.numpy()), v_is_view=np.int32(view), block_table=perm.numpy(),
                            kv_seq_lens=lens.numpy(), sm_scale=np.float32(sm_scale), page_size=np.int32(page), out=out.numpy())
        print(name, "ok", float(out.abs().max()))


if __name__ == "__main__":
    main()
