cd "$(dirname "$0")/.."
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dec2_dec3_fused" 2>&1 | tail -5 || exit 1
for tag in base ${TAGS}; do
  echo "== $tag"
  if [ $tag = base ]; then unset FLDR_LIB; else export FLDR_LIB=tools/stamps/libfldr_$tag.so; fi
  timeout -k 10 200 python tools/dec23_probe.py 2>&1 | grep -E "dec2|wave" | tail -3 || exit 1
done
