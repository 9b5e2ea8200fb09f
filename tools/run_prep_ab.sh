cd "$(dirname "$0")/.."
for tag in base ${TAGS:-prepab1 prepab2 prepab3} base; do
  echo "== $tag"
  if [ $tag = base ]; then unset FLDR_LIB; else export FLDR_LIB=tools/stamps/libfldr_$tag.so; fi
  timeout -k 10 200 python tools/kernel_bench.py prep 2>&1 | grep -E "level0_prep|phase" || exit 1
done
