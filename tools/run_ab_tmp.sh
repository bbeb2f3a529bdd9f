python -m pytest tests/test_gpu_dtw_features.py -m gpu -x -q -k "dtw" 2>&1 | tail -2
for i in 1 2; do
for ng in 1 2; do echo "NG=$ng"; ABN_DTW_NG=$ng python tools/dtw_time.py 2>&1 | grep "dtw 10000"; done
done
ABN_DTW_NG=2 ABN_DTW_WGS=4 python tools/dtw_time.py 2>&1 | grep "dtw 10000"
ABN_DTW_NG=2 tools/prof.sh dtwng2 tools/dtw_kernels.py 2>&1 | head -5
