// Host-side AddressSanitizer run of the C ABI: the library's host code compiled with
// -fsanitize=address (host pass only: GPU ASan is not available on this pool) into ONE executable with this driver, which
// walks every entry point that needs no GPU -- descriptor checks, the parameter tables of both network kinds for every
// golden shape, name-buffer truncation, null / bad arguments of the device entry points (they must fail cleanly before
// touching the device) and context creation without a device.  Built and run by `make -C csrc asan`.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/trebi_hip.h"

static int fails = 0;
#define EXPECT(cond)                                                        \
  do {                                                                      \
    if (!(cond)) { ++fails; fprintf(stderr, "FAILED %s:%d: %s (%s)\n", __FILE__, __LINE__, #cond, trebi_last_error()); } \
  } while (0)

int main() {
  EXPECT(trebi_version() >= 1);
  EXPECT(trebi_last_error() != nullptr);
  const int mults[][8] = {{1, 2, 4, 8}, {1, 4, 8}, {1, 2, 4, 8, 16, 32}};
  const int nm[] = {4, 3, 6};
  const int shapes[][3] = {{32, 4, 0}, {32, 14, 0}, {32, 23, 0}, {4, 23, 1}, {128, 4, 2}};   // horizon, transition_dim, mults row
  long long total = 0;
  for (int kind = 0; kind < 2; ++kind)
    for (auto& sh : shapes) {
      trebi_net_desc d;
      memset(&d, 0, sizeof d);
      d.kind = kind; d.horizon = sh[0]; d.transition_dim = sh[1]; d.dim = 32; d.max_timesteps = 20;
      d.n_mults = nm[sh[2]];
      for (int i = 0; i < d.n_mults; ++i) d.dim_mults[i] = mults[sh[2]][i];
      const int n = trebi_net_param_count(&d);
      EXPECT(n > 0);
      for (int i = 0; i < n; ++i) {
        char name[96], tiny[5];
        int64_t numel = 0;
        EXPECT(trebi_net_param_info(&d, i, name, sizeof name, &numel) == 0);
        EXPECT(numel > 0 && strlen(name) > 0);
        EXPECT(trebi_net_param_info(&d, i, tiny, sizeof tiny, nullptr) == 0);      // truncation, no overrun
        EXPECT(strlen(tiny) <= 4);
        total += numel;
      }
      EXPECT(trebi_net_param_info(&d, n, nullptr, 0, nullptr) != 0);
      EXPECT(trebi_net_param_info(&d, -1, nullptr, 0, nullptr) != 0);
    }
  printf("walked %lld parameters\n", total);
  {   // descriptors the library must refuse
    trebi_net_desc d;
    memset(&d, 0, sizeof d);
    EXPECT(trebi_net_param_count(nullptr) < 0);
    d.kind = 2;
    EXPECT(trebi_net_param_count(&d) < 0);
    d.kind = 0; d.n_mults = 9;
    EXPECT(trebi_net_param_count(&d) < 0);
    d.n_mults = 2; d.dim_mults[0] = 1; d.dim_mults[1] = 2; d.horizon = 8; d.transition_dim = 4; d.max_timesteps = 20; d.dim = 24;
    EXPECT(trebi_net_param_count(&d) == TREBI_UNSUPPORTED_CONFIG);
  }
  // device entry points with null / bad arguments: clean error codes, nothing dereferenced
  trebi_ctx* ctx = nullptr;
  EXPECT(trebi_ctx_create(0, nullptr) != 0);
  const int rc = trebi_ctx_create(0, &ctx);          // no GPU in the build container: an error code, not a crash
  printf("trebi_ctx_create(0): %d (%s)\n", rc, rc ? trebi_last_error() : "a device is present");
  EXPECT(trebi_net_create(nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr) != 0);
  EXPECT(trebi_unet_forward(nullptr, nullptr, 0, nullptr, nullptr, 0, 0, nullptr, 0, nullptr) != 0);
  EXPECT(trebi_unet_forward_static(nullptr, nullptr, 0, nullptr, nullptr, 0, 0, nullptr) != 0);
  EXPECT(trebi_value_gradients(nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0, 0, nullptr, 0, nullptr) != 0);
  EXPECT(trebi_sampler_create(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr) != 0);
  EXPECT(trebi_sampler_bind(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) != 0);
  EXPECT(trebi_sampler_run(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) != 0);
  EXPECT(trebi_trainer_create(nullptr, nullptr, 0, nullptr) != 0);
  EXPECT(trebi_unet_workspace_bytes(nullptr, 4) == 0);
  EXPECT(trebi_value_workspace_bytes(nullptr, 4) == 0);
  EXPECT(trebi_net_flops(nullptr, 0) == 0.0);
  trebi_net_destroy(nullptr);
  trebi_sampler_destroy(nullptr);
  trebi_trainer_destroy(nullptr);
  if (ctx) trebi_ctx_destroy(ctx);
  printf(fails ? "asan host run: %d FAILED\n" : "asan host run: ok (%d failures)\n", fails);
  return fails ? 1 : 0;
}
