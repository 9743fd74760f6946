echo base; python tools/gemm_ab.py
for m in 1 2; do for d in 1 2 4; do echo "mode $m delay $d"; BTR_GEMM_DMODE=$m BTR_GEMM_DN=$d python tools/gemm_ab.py; done; done
