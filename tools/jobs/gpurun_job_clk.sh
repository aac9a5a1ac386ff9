cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CU_CYCLES --kernel-trace -d $R/gpurun_out/clk -o c -- python $R/tools/sq_probe.py conv2d gemm conv3d > $R/gpurun_out/clk.log 2>&1
cd $R
python - <<'PY'
import sqlite3, collections
con = sqlite3.connect("gpurun_out/clk/c_results.db")
cols = [d[0] for d in con.execute("select * from counters_collection limit 1").description]
print(cols)
name_col = "kernel_name" if "kernel_name" in cols else "name"
rows = con.execute("select %s, counter_name, avg(value) from counters_collection group by %s, counter_name" % (name_col, name_col)).fetchall()
dur = dict(con.execute("select name, avg(duration) from kernels group by name").fetchall())
agg = collections.defaultdict(dict)
for n, c, v in rows: agg[n][c] = v
for n, d in agg.items():
    if not any(s in n for s in ("conv2d_kernel", "gemm_direct", "conv3d_k3")): continue
    us = dur.get(n, 0) / 1e3
    print("%-70s %7.1f us  GUI_ACTIVE %.3g -> %.2f GHz   waves %.0f busy_cu %.3g" % (n[:70], us, d.get("GRBM_GUI_ACTIVE", 0), d.get("GRBM_GUI_ACTIVE", 0) / max(us, 1e-9) / 1e3, d.get("SQ_WAVES", 0), d.get("SQ_BUSY_CU_CYCLES", 0)))
PY
rm -rf gpurun_out/clk
