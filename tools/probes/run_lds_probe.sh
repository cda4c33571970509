cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
$R/tools/probes/lds_probe
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_MFMA -d /tmp/lp -- $R/tools/probes/lds_probe > /dev/null 2>&1
DB=$(find /tmp/lp -name "*.db" | head -1)
python - "$DB" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name").fetchall()
d = {}
for k, c, v in rows:
    d.setdefault(k, {})[c] = v
for k, v in d.items():
    print(k[:40], {c: f"{x:.4g}" for c, x in v.items()})
PY
