// placeholder, replaced below
#include "pmf_device.h"
extern "C" int pmf_gauss_factor_sweep(pmf_ctx *, int, double, double) { pmf_set_error("not built yet"); return PMF_EINVAL; }
extern "C" int pmf_gauss_bias_sweep(pmf_ctx *, int, double, double) { pmf_set_error("not built yet"); return PMF_EINVAL; }
extern "C" int pmf_gauss_factor_accumulate(pmf_ctx *, int, void *) { pmf_set_error("not built yet"); return PMF_EINVAL; }
extern "C" int pmf_gauss_factor_finalize(pmf_ctx *, int, const void *, double, double) { pmf_set_error("not built yet"); return PMF_EINVAL; }
extern "C" int pmf_gauss_bias_accumulate(pmf_ctx *, int, void *) { pmf_set_error("not built yet"); return PMF_EINVAL; }
extern "C" int pmf_gauss_bias_finalize(pmf_ctx *, int, const void *, double, double) { pmf_set_error("not built yet"); return PMF_EINVAL; }
extern "C" int pmf_topk_items(pmf_ctx *, int64_t, const int32_t *, int, int, int32_t *, double *) { pmf_set_error("not built yet"); return PMF_EINVAL; }
