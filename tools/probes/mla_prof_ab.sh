cd /tmp && export TMPDIR=/tmp
for w8 in 1 0; do
MI_MLA_WIDE8=$w8 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_w8_$w8 -o out -- python -c "
import os, sys, torch
sys.path.insert(0, '$GRAFT_REPO_ROOT/sgl-kernel-npu_amd/python')
from sgl_kernel_npu.bench_hooks import _mla_inputs
import sgl_kernel_npu
q, kn, kr, bt, lens = _mla_inputs(128, 128, 4096, 64)
out = torch.empty((128, 128, 512), dtype=torch.bfloat16, device='cuda')
for _ in range(200): torch.ops.npu.decode_mla(q, kn, kr, out, lens, 576 ** -0.5, 64, bt, 0)
torch.cuda.synchronize()
" > /dev/null 2>&1
echo "WIDE8=$w8"; find /tmp/prof_w8_$w8 -name "*kernel_stats.csv" | head -1 | xargs head -5
done
