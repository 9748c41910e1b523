// Entry points not implemented yet: they fail loudly.
#include "sa_internal.h"
extern "C" {
int sa_align_batch_host(const char* const*, const uint32_t*, const char* const*, const uint32_t*, size_t, int, int, int, int, int, int32_t*, uint32_t*, char**) { return sa::fail(SA_ERR_ARGUMENT, "sa_align_batch_host: not implemented"); }
void sa_free(void* p) { free(p); }
int sa_minimize_batch_host(const char* const*, const uint32_t*, size_t, uint32_t, uint32_t, int, int, uint32_t**, uint64_t*) { return sa::fail(SA_ERR_ARGUMENT, "not implemented"); }
int sa_remove_duplicate_host(uint32_t*, size_t, uint32_t*, size_t*) { return sa::fail(SA_ERR_ARGUMENT, "not implemented"); }
}
