// placeholder until the mel front-end lands (keeps the ABI symbol set complete)
#include "common.h"
using namespace dsvc;
extern "C" {
int dsvc_melspec_create(const dsvc_melspec_cfg*, const float*, dsvc_melspec**) { return fail(DSVC_ESTATE, "melspec not built yet"); }
void dsvc_melspec_destroy(dsvc_melspec*) {}
int dsvc_melspec_frames(const dsvc_melspec*, int64_t, int32_t*) { return fail(DSVC_ESTATE, "melspec not built yet"); }
int dsvc_melspec_run(dsvc_melspec*, const float*, float*, int32_t, int64_t, void*) { return fail(DSVC_ESTATE, "melspec not built yet"); }
}
