export PYTHONPATH=.
for cfg in "16 0" "4 0" "4 1" "2 1"; do
  set -- $cfg
  echo "== GEMV_MAX_ROWS=$1 FOLD_NORM=$2"
  BIGDL_GEMV_MAX_ROWS=$1 BIGDL_FOLD_NORM=$2 timeout -k 10 200 python benchmark/batch_decode_probe.py --batch 4,8,16 --ctx 1024 --steps 32 2>/dev/null | grep ms_per_step
done
