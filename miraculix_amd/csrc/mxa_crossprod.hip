// placeholder, replaced below
#include "../../include/miraculix_amd.h"
#include "mxa_internal.h"
extern "C" int snp_multiply_gpu(unsigned char *, int, int, double *, bool) { mxa::set_error(99, "snp_multiply_gpu: not built yet"); return 1; }
