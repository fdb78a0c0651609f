"""Per-kernel register / scratch / instruction-mix statistics of one source file compiled to gfx950 assembly.
    python tools/kstat.py ngm_field_bwd_b3.hip [-DFLAG ...] [--filter=substr]
"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flags = [a for a in sys.argv[2:] if not a.startswith("--filter=")]
filt = [a.split("=", 1)[1] for a in sys.argv[2:] if a.startswith("--filter=")]
path = src if os.path.exists(src) else os.path.join(ROOT, "neural_graph_mapping_amd", "csrc", src)
out = tempfile.mktemp(suffix=".s")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Wno-unused-result",
                "-I" + os.path.join(ROOT, "include")] + flags + [path, "-o", out], check=True, stderr=subprocess.DEVNULL)
txt = open(out).read()
os.remove(out)
# split into functions
funcs = re.split(r"\n(?=\s*\.globl\s)", txt)
for f in funcs:
    m = re.search(r"\.globl\s+(\S+)", f)
    if not m or ".amdhsa_kernel" not in f:
        continue
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    if filt and not any(s in name for s in filt):
        continue
    def g(k):
        mm = re.search(r"\." + k + r"\s+(\S+)", f)
        return mm.group(1) if mm else "?"
    cnt = lambda op: len(re.findall(r"^\s+" + op, f, re.M))
    stats = dict(mfma=cnt("v_mfma"), valu=cnt("v_(?!mfma)"), dot2c=cnt("v_dot2c"), perm=cnt("v_perm"), pk=cnt("v_pk_"),
                 ds_read=cnt("ds_read"), ds_write=cnt("ds_write"), scratch_ld=cnt("scratch_load"), scratch_st=cnt("scratch_store"),
                 s_nop=cnt("s_nop"), dma=cnt("global_load_lds"))
    print(f"{name[:110]}\n   vgpr {g('amdhsa_next_free_vgpr')} accum_offset {g('amdhsa_accum_offset')} sgpr {g('amdhsa_next_free_sgpr')} "
          f"scratch {g('amdhsa_private_segment_fixed_size')} | " + " ".join(f"{k} {v}" for k, v in stats.items()))
