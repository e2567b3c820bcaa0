// Single translation unit of libsgmse_hip.so:  hipcc --offload-arch=gfx950 -O3 -shared -fPIC sgmse_hip.hip
#include "capi.h"
