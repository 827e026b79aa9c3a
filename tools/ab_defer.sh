for dv in 16 32 64; do export EMG_DEFER=$dv; echo defer=$dv; bash tools/bench_others.sh C3 C3z C1 C5; done
