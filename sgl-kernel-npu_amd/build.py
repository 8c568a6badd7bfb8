"""Build driver for the MI355X (gfx950) native code.  No cmake: hipcc / g++ directly, in-tree outputs
under sgl-kernel-npu_amd/lib/ (git-ignored, shipped to the GPU box by gpurun).

  libmi_ep.so            HIP kernels + C-ABI of include/mi_ep.h         (hipcc, no torch)
  libmi_sgl_kernels.so   HIP kernels + C-ABI of include/mi_sgl_kernels.h (hipcc, no torch)
  deep_ep_cpp*.so        pybind11 host runtime (deep_ep.Buffer backend)  (g++ + torch headers)
  libsgl_kernel_npu.so   torch.ops.npu.* registrations                   (g++ + torch headers)
"""
import glob
import os
import shutil
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "lib")
INC = os.path.join(ROOT, "include")
ARCH = "gfx950"


def _newer(srcs, out):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(s) > t for s in srcs)


def _run(cmd):
    print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _file_flags(src):
    """Per-file compiler flags: an optional first line `// hipcc-flags: ...` in the source."""
    with open(src) as f:
        first = f.readline()
    return first.split(":", 1)[1].split() if first.startswith("// hipcc-flags:") else []


def build_hip_lib(name, subdir, extra=()):
    """Each .hip is compiled to an object on its own (in parallel, with its own flags), then linked."""
    os.makedirs(LIB, exist_ok=True)
    objdir = os.path.join(LIB, "obj", subdir)
    os.makedirs(objdir, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(HERE, "csrc", subdir, "*.hip")))
    hdrs = glob.glob(os.path.join(HERE, "csrc", subdir, "*.h")) + glob.glob(os.path.join(INC, "*.h")) + glob.glob(os.path.join(HERE, "csrc", "*.h"))
    out = os.path.join(LIB, name)
    common = [hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-strict-aliasing", f"-I{INC}",
              f"-I{os.path.join(HERE, 'csrc', subdir)}", f"-I{os.path.join(HERE, 'csrc')}", *extra]
    objs, procs = [], []
    for src in srcs:
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if _newer([src] + hdrs, obj):
            cmd = common + _file_flags(src) + ["-c", src, "-o", obj]
            print("[build]", " ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    if srcs and _newer(objs, out):
        _run([hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", out])
    return out


def _torch_flags():
    import torch
    from torch.utils import cpp_extension as ce

    tdir = os.path.dirname(torch.__file__)
    inc = [f"-I{p}" for p in ce.include_paths()] + ["-I/opt/rocm/include", f"-I{sysconfig.get_paths()['include']}"]
    libdir = os.path.join(tdir, "lib")
    cxx11 = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    defs = ["-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", f"-D_GLIBCXX_USE_CXX11_ABI={cxx11}"]
    link = [f"-L{libdir}", f"-Wl,-rpath,{libdir}", "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_hip", "-lc10_hip",
            "-lamdhip64", f"-L{LIB}", "-Wl,-rpath,$ORIGIN"]
    return inc, defs, link


def build_host_ext(out_name, srcs, libs, python_module=False):
    os.makedirs(LIB, exist_ok=True)
    srcs = [os.path.join(HERE, "csrc", s) for s in srcs]
    deps = srcs + glob.glob(os.path.join(INC, "*.h")) + glob.glob(os.path.join(HERE, "csrc", "**", "*.h*"), recursive=True)
    out = os.path.join(LIB, out_name)
    if not _newer(deps, out):
        return out
    inc, defs, link = _torch_flags()
    extra = []
    if python_module:
        import pybind11

        inc.append(f"-I{pybind11.get_include()}")
        link.append("-ltorch_python")
        defs.append(f"-DTORCH_EXTENSION_NAME={out_name.split('.')[0]}")
    _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-deprecated-declarations", "-Wno-unused-result", f"-I{INC}",
          f"-I{os.path.join(HERE, 'csrc')}", *inc, *defs, *srcs, "-o", out, *link, *[f"-l{l}" for l in libs], *extra])
    return out


def ext_suffix():
    return sysconfig.get_config_var("EXT_SUFFIX") or ".so"


def build_all(verbose=True):
    outs = [build_hip_lib("libmi_ep.so", "ep")]
    if glob.glob(os.path.join(HERE, "csrc", "kernels", "*.hip")):
        outs.append(build_hip_lib("libmi_sgl_kernels.so", "kernels"))
    if os.path.exists(os.path.join(HERE, "csrc", "deepep", "deep_ep.cpp")):
        outs.append(build_host_ext("deep_ep_cpp" + ext_suffix(),
                                   ["deepep/deep_ep.cpp", "deepep/pybind_extension.cpp"], ["mi_ep"], python_module=True))
    if os.path.exists(os.path.join(HERE, "csrc", "pytorch_extensions.cpp")):
        outs.append(build_host_ext("libsgl_kernel_npu.so", ["pytorch_extensions.cpp"], ["mi_sgl_kernels"]))
    return outs


if __name__ == "__main__":
    for o in build_all():
        print("built", o)
