"""Every distinct convolution of resnet50_gru.json at 16 clips (spec/RESNET50_BLOCK.md; M = 16 * 600 * W rows) through the kernel entry points,
each called twice (the second launch is the warm one).  (The model runs the 3x3 of stages 0 and 1 on the conv blocks' implicit-GEMM kernels
instead — api.hip: rn_c1_direct —, so the s0.c1 / s1.c1 rows here show the product path those stages no longer take by default.)  Run under `rocprofv3 --kernel-trace --output-format csv` and feed the trace to this
script's `report` mode for per-shape kernel times:
    rocprofv3 --kernel-trace --output-format csv -d OUT -o rn -- python3 tools/tune_rn_products.py run
    python3 tools/tune_rn_products.py report OUT/.../rn_kernel_trace.csv"""
import csv, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
B, H = 16, 600
SHAPES = []      # (tag, W, Cin, Cout, ksize, stride_f)
cin, W = 64, 16
for s, (w, nb) in enumerate(zip((32, 64, 128, 256), (3, 4, 6, 3))):
    st = 2 if s > 0 else 1
    SHAPES += [(f"s{s}.c0first", W, cin, w, 1, st), (f"s{s}.sc", W, cin, 4 * w, 1, st)]
    W //= st
    SHAPES += [(f"s{s}.c0", W, 4 * w, w, 1, 1), (f"s{s}.c1", W, w, w, 3, 1), (f"s{s}.c2", W, w, 4 * w, 1, 1)]
    cin = 4 * w

if sys.argv[1] == "run":
    import ctypes as C, torch
    from seld_amd import _lib
    lib = _lib.load()
    if os.environ.get("SELD_GSB_DBG"):
        assert lib.seld_k_set_option(b"gsb_dbg", int(os.environ["SELD_GSB_DBG"])) == 0      # 8: force the 16-wave form, 4: the 4-wave form
    p = lambda t: C.c_void_p(t.data_ptr())
    for tag, W, Cin, Cout, k, st in SHAPES:
        x = torch.randn(B, H, W, Cin, device="cuda"); w = torch.randn(k, k, Cin, Cout, device="cuda") * 0.05
        z = torch.empty(B, H, W // st, Cout, device="cuda"); dz = torch.randn_like(z)
        dw = torch.empty_like(w); dx = torch.empty_like(x)
        for _ in range(2):
            assert lib.seld_k_rn_conv(p(x), p(w), p(z), B, H, W, Cin, Cout, k, st) == 0
            assert lib.seld_k_rn_conv_bwd(p(x), p(w), p(dz), p(dw), p(dx), B, H, W, Cin, Cout, k, st) == 0
        del x, w, z, dz, dw, dx
    sys.exit(0)

rows = list(csv.DictReader(open(sys.argv[2])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
keep = [r for r in rows if any(t in r["Kernel_Name"] for t in ("gemm_", "reduce_slabs"))]
# per shape and call: fwd product; bwd: wgrad product (+ reduce_slabs2), dgrad product; split launches precede their product
i = 0
def take(pred, most=1 << 30):
    global i
    out = []
    while i < len(keep) and len(out) < most and pred(keep[i]["Kernel_Name"]):
        out.append(keep[i]); i += 1
    return out
us = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
short = lambda n: n.split("(")[0].replace("void ", "")
tot = {"fwd": 0.0, "wgrad": 0.0, "dgrad": 0.0}
mult = {"c0first": 1, "sc": 1, "c0": None, "c1": None, "c2": None}
nb = {"s0": 3, "s1": 4, "s2": 6, "s3": 3}
for tag, W, Cin, Cout, k, st in SHAPES:
    M, K = B * H * (W // st), k * k * Cin
    rec = {}
    for rep in range(2):
        take(lambda n: "split_b" in n)
        f = take(lambda n: "split_b" not in n and "gemm_tn" not in n and "reduce_slabs" not in n, 1)
        take(lambda n: "split_b" in n)
        wg = take(lambda n: "gemm_tn" in n); rs = take(lambda n: "reduce_slabs" in n)
        take(lambda n: "split_b" in n)
        dg = take(lambda n: "split_b" not in n and "gemm_tn" not in n and "reduce_slabs" not in n, 1)
        rec = dict(fwd=f, wgrad=wg + rs, dgrad=dg)
    flop = 2.0 * M * K * Cout
    stage, kind = tag.split(".")
    n = 1 if kind in ("c0first", "sc") else (nb[stage] - (1 if kind == "c0" else 0))
    line = f"{tag:11s} M={M:6d} K={K:4d} N={Cout:4d} x{n}:"
    for what in ("fwd", "wgrad", "dgrad"):
        t = sum(us(r) for r in rec[what])
        tot[what] += n * t
        line += f"  {what} {short(rec[what][0]['Kernel_Name'])[:18]:18s} {t:7.1f} us {flop / t / 1e6:6.1f} TF"
    print(line)
print("per step (all blocks): " + "  ".join(f"{k} {v / 1e3:.2f} ms" for k, v in tot.items()))
