cd "$(dirname "$0")/.."
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "dec2_dec3 or golden or forward or fullsize or strong or graphed" 2>&1 | tail -5 || exit 1
timeout -k 10 200 python tools/dec23_probe.py 2>&1 | grep -E "dec2|wave" | tail -2
