python -m pytest tests -m gpu -x -q -k "row_outliers or attention_vjp" 2>&1 | tail -n 6
python - <<'PY'
import json; m=json.load(open('gpurun_out/tolerances_measured.json')); print({k:v['max'] for k,v in m.items() if 'rows' in k})
PY
