set -o pipefail
R=/root/repo
cd $R/image-classification-xai_amd
python - <<'PY'
import torch
from xai_engine.selfcheck import streams_probe
print("probe alone, fresh process:", streams_probe(torch.device("cuda:0")))
print("probe again:", streams_probe(torch.device("cuda:0")))
PY
timeout -k 10 300 python -m xai_engine.selfcheck 2>&1 | tail -3
