// Internal (not part of the C ABI): hand-over from adm_conv's dispatcher to the resident-tile 1x1 kernel.
#pragma once
#include "adm_hip.h"

// pixels per resident tile (128 or 64) if these arguments run on the resident-tile 1x1 kernel, else 0
int adm_conv1x1_resident_bm(const adm_conv_args* a);
// launch it (arguments already validated by adm_conv); out_stats, if given, has H*W / bm slabs per image
int adm_conv1x1_resident_launch(const adm_conv_args* a, int bm, void* stream);
