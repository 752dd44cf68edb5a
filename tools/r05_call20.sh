export CMDI_LIB_VARIANT=oldprob
python -m pytest tests -m gpu -q -k "row_outliers" 2>&1 | grep -E "assert|AssertionError|passed|failed|worst" | head -8
