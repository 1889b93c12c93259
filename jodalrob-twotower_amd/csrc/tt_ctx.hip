// Context, error string, ABI version.
#include "tt_common.h"

#include <string.h>

static thread_local char g_err[512] = "";

void tt_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" {

int tt_abi_version(void) { return TT_ABI_VERSION; }

const char* tt_last_error_string(void) { return g_err; }

int tt_ctx_create(int device, tt_ctx** out) {
  TT_CHECK_ARG(out != nullptr, "tt_ctx_create: out is NULL");
  int n = 0;
  TT_HIP(hipGetDeviceCount(&n));
  TT_CHECK_ARG(device >= 0 && device < n, "tt_ctx_create: device %d out of range (%d devices)", device, n);
  hipDeviceProp_t prop;
  TT_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    tt_set_error("tt_ctx_create: device %d is %s; this library is built for gfx950 (MI355X) only", device,
                 prop.gcnArchName);
    return TT_ERR_UNSUPPORTED;
  }
  tt_ctx* c = new tt_ctx();
  c->device = device;
  c->num_cus = prop.multiProcessorCount;
  c->lds_per_block = prop.sharedMemPerBlock;
  c->lookup_stamps = nullptr;
  c->lookup_stamp_slots = 0;
  *out = c;
  return TT_OK;
}

int tt_ctx_destroy(tt_ctx* ctx) {
  delete ctx;
  return TT_OK;
}

int tt_ctx_num_cus(const tt_ctx* ctx) { return ctx ? ctx->num_cus : 0; }

}  // extern "C"
