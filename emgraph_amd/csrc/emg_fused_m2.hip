// the fused in-place training kernels of model 2 (include/emgraph_hip.h: EMG_TRANSE_L1 .. EMG_HOLE), a translation unit of
// its own so that the five models compile in parallel (emg_fused_inst.inc)
#define EMG_FUSED_MODEL 2
#include "emg_fused_inst.inc"
