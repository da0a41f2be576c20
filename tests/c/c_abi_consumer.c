/* A plain-C consumer of include/wd_hip.h: what a cgo / JNI / N-API binding sees.  Built and run by
 * tests/test_c_abi_and_host.py on the CPU-only container: it links nothing but libdl, opens
 * libwdhip.so, resolves entry points through the header's own prototypes and checks the error
 * convention (integer return code + wd_last_error) when no device is usable. */
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include "wd_hip.h"

#define RESOLVE(name) \
  __typeof__(name) *p_##name = (__typeof__(name) *)dlsym(lib, #name); \
  if (!p_##name) { fprintf(stderr, "missing %s\n", #name); return 2; }

int main(int argc, char **argv) {
  if (argc < 2) return 64;
  void *lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!lib) { fprintf(stderr, "%s\n", dlerror()); return 1; }
  RESOLVE(wd_version)
  RESOLVE(wd_last_error)
  RESOLVE(wd_init)
  RESOLVE(wd_malloc)
  RESOLVE(wd_plan_create)
  RESOLVE(wd_launch_packed)
  const char *version = p_wd_version();
  if (!version || strlen(version) == 0) return 3;
  /* without a HIP runtime / device every compute entry point must fail loudly, never fall back */
  void *dptr = (void *)0;
  int rc_init = p_wd_init(0);
  int rc_malloc = p_wd_malloc(1024, &dptr);
  const char *msg = p_wd_last_error();
  printf("version %s init=%d malloc=%d last_error=\"%s\"\n", version, rc_init, rc_malloc, msg ? msg : "");
  if (rc_init == 0 && rc_malloc == 0) return 0;          /* a GPU box: fine */
  if (rc_malloc == 0 || !msg || strlen(msg) == 0) return 4; /* failure must carry a message */
  return 0;
}
