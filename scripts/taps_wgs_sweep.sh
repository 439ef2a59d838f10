for w in 1 2 3 4; do echo "== SR_TAPS_WGS=$w"; SR_TAPS_WGS=$w SR_CONVT_FUSED=0 SR_CONVT_TAPS=1 python scripts/bench_convt_small.py child 2>&1 | grep -v amdgpu | grep "res32\|res16\|res8\|res4"; done
