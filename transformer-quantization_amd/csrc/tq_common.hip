// Error reporting + argument validation for the C ABI (no kernels here).
#include <string.h>

#include "tq_host.h"

namespace tq {

static thread_local char g_err[512] = "";

int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int check_quantizer(const tq_quantizer* q, uint64_t n, const char* who) {
  if (q == nullptr) return set_error(TQ_EINVAL, "%s: quantizer is NULL", who);
  if (q->delta == nullptr) return set_error(TQ_EINVAL, "%s: quantizer.delta is NULL (not initialised)", who);
  if (!q->symmetric && q->zero_float == nullptr)
    return set_error(TQ_EINVAL, "%s: asymmetric quantizer needs zero_float", who);
  if (q->n_bits < 1 || q->n_bits > 24) return set_error(TQ_EINVAL, "%s: n_bits=%d outside 1..24", who, q->n_bits);
  if (q->n_params == 0) return set_error(TQ_EINVAL, "%s: n_params == 0", who);
  if (q->n_params > 1) {
    if (q->inner == 0) return set_error(TQ_EINVAL, "%s: inner == 0", who);
    if (n % (q->n_params * q->inner) != 0)
      return set_error(TQ_EINVAL, "%s: n=%llu is not a multiple of n_params*inner=%llu", who,
                       (unsigned long long)n, (unsigned long long)(q->n_params * q->inner));
  }
  return TQ_OK;
}

}  // namespace tq

extern "C" {

int tq_abi_version(void) { return TQ_ABI_VERSION; }

const char* tq_last_error(void) { return tq::g_err; }

}  // extern "C"
