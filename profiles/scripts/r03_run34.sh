set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3ai; mkdir -p $out
( while true; do echo "[heartbeat $(date +%T)]"; sleep 60; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
cd $R
b() { timeout -k 10 300 python bench.py --lean --steps 5 > $out/$1.json 2> $out/$1.err; python3 -c "import json;d=json.load(open('$out/$1.json'));print('$1', round(d['value'],2), 'K2 frac', round(d['roofline']['frac'],3))"; }
b A_cold; b B_second
wc -l ~/.config/miopen/*.ufdb.txt
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -m gpu -q > $out/pytest.txt 2>&1; tail -1 $out/pytest.txt
wc -l ~/.config/miopen/*.ufdb.txt
b C_after_config_tests
cp ~/.config/miopen/*.ufdb.txt $out/ufdb_after_tests.txt
MIOPEN_USER_DB_PATH=$(mktemp -d) b D_private_empty_db
b E_default_again
