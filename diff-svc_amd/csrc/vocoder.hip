// placeholder until the vocoder lands (keeps the ABI symbol set complete)
#include "common.h"
using namespace dsvc;
extern "C" {
int dsvc_vocoder_create(const dsvc_vocoder_cfg*, dsvc_vocoder**) { return fail(DSVC_ESTATE, "vocoder not built yet"); }
int dsvc_vocoder_load_tensor(dsvc_vocoder*, const char*, const float*, int64_t) { return fail(DSVC_ESTATE, "vocoder not built yet"); }
int dsvc_vocoder_finalize(dsvc_vocoder*) { return fail(DSVC_ESTATE, "vocoder not built yet"); }
void dsvc_vocoder_destroy(dsvc_vocoder*) {}
int dsvc_vocode(dsvc_vocoder*, const float*, const float*, float*, int32_t, int32_t, uint64_t, int32_t, void*) { return fail(DSVC_ESTATE, "vocoder not built yet"); }
}
