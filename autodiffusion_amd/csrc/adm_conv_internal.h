// Internal (not part of the C ABI): hand-over from adm_conv's dispatcher to the resident-tile 1x1 kernel.
#pragma once
#include "adm_hip.h"

// pixels per resident tile (128 or 64) if these arguments run on the resident-tile 1x1 kernel, else 0;
// *wm (optional) = wave rows: 1 -> 384-wide Cout blocks, 2 -> 192-wide (narrow outputs)
int adm_conv1x1_resident_cfg(const adm_conv_args* a, int* wm);
// out_stats slabs per image on that kernel (0 if it does not take these arguments)
int adm_conv1x1_resident_slabs(const adm_conv_args* a);
// launch it (arguments already validated by adm_conv)
int adm_conv1x1_resident_launch(const adm_conv_args* a, void* stream);
