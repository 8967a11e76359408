/* Plain-C client of include/poco_hip.h: proves the boundary needs nothing but a C compiler and dlopen.
 * No GPU: it only uses the host-side part of the ABI (declarations of a variant, strict loading errors).
 * usage: abi_check <path to libpoco_hip.so>;  prints "ok <n tensors>" */
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include "poco_hip.h"

typedef const char* (*last_error_fn)(void);
typedef int (*create_fn)(const char*, int, int, poco_handle_t*);
typedef int (*create_ex_fn)(const char*, int, int, const char*, poco_handle_t*);
typedef int (*version_fn)(void);
typedef void (*destroy_fn)(poco_handle_t);
typedef int (*num_tensors_fn)(poco_handle_t);
typedef int (*tensor_info_fn)(poco_handle_t, int, char*, size_t, int64_t*, int*, int*);
typedef int (*load_tensor_fn)(poco_handle_t, const char*, const float*, const int64_t*, int);
typedef int (*forward_fn)(poco_handle_t, int, const poco_inputs_t*, const poco_outputs_t*, void*);
typedef int (*tune_fn)(int, int, int, int, int, int, int, const int*, int, int, float*, void*);
typedef int (*status_fn)(poco_handle_t);

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  void* so = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!so) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 3; }
  last_error_fn last_error = (last_error_fn)dlsym(so, "poco_last_error");
  create_fn create = (create_fn)dlsym(so, "poco_create");
  destroy_fn destroy = (destroy_fn)dlsym(so, "poco_destroy");
  num_tensors_fn num_tensors = (num_tensors_fn)dlsym(so, "poco_num_tensors");
  tensor_info_fn tensor_info = (tensor_info_fn)dlsym(so, "poco_tensor_info");
  load_tensor_fn load_tensor = (load_tensor_fn)dlsym(so, "poco_load_tensor");
  forward_fn forward = (forward_fn)dlsym(so, "poco_forward");
  create_ex_fn create_ex = (create_ex_fn)dlsym(so, "poco_create_ex");
  version_fn version = (version_fn)dlsym(so, "poco_abi_version");
  tune_fn tune = (tune_fn)dlsym(so, "poco_tune_conv");
  status_fn status = (status_fn)dlsym(so, "poco_status");
  if (!tune || !status) return 4;
  if (!last_error || !create || !destroy || !num_tensors || !tensor_info || !load_tensor || !forward || !create_ex || !version) return 4;
  if (version() != POCO_ABI_VERSION) { fprintf(stderr, "library ABI %d, header ABI %d\n", version(), POCO_ABI_VERSION); return 15; }

  poco_handle_t h = 0;
  if (create("no_such-variant", 4, 1, &h) == 0) return 5;              /* unknown variant is an error ... */
  if (strlen(last_error()) == 0) return 6;                             /* ... with a message */
  if (create("resnet50-cliff", 4, 1, &h) != 0) { fprintf(stderr, "%s\n", last_error()); return 7; }
  int n = num_tensors(h);
  if (n < 100) return 8;
  int found = 0;
  for (int i = 0; i < n; ++i) {
    char name[256]; int64_t shape[8]; int rank = 0, required = 0;
    if (tensor_info(h, i, name, sizeof name, shape, &rank, &required) != 0) return 9;
    if (strcmp(name, "backbone.conv1.weight") == 0) {
      found = 1;
      if (rank != 4 || shape[0] != 64 || shape[1] != 3 || shape[2] != 7 || shape[3] != 7 || !required) return 10;
    }
  }
  if (!found) return 11;
  float w[4] = {0};
  int64_t bad_shape[1] = {4};
  if (load_tensor(h, "backbone.conv1.weight", w, bad_shape, 1) == 0) return 12;     /* strict shapes */
  if (load_tensor(h, "backbone.nope", w, bad_shape, 1) == 0) return 13;             /* strict names */
  poco_inputs_t in; poco_outputs_t out;
  memset(&in, 0, sizeof in); memset(&out, 0, sizeof out);
  if (forward(h, 1, &in, &out, 0) != 1 || !strstr(last_error(), "struct_size")) return 20;   /* no size word: POCO_ERR_ARG, nothing else read */
  in.struct_size = sizeof in; out.struct_size = sizeof out;
  if (forward(h, 1, &in, &out, 0) != 3 || !strstr(last_error(), "finalize")) return 14;      /* well-formed, but not finalized: POCO_ERR_STATE */
  out.struct_size = sizeof out - 4;                                                           /* a truncated struct (not pointer-granular) */
  if (forward(h, 1, &in, &out, 0) != 1 || !strstr(last_error(), "poco_outputs_t.struct_size")) return 21;
  out.struct_size = sizeof out - sizeof(float*);                                              /* a binding from a header without `record`: accepted */
  if (forward(h, 1, &in, &out, 0) != 3) return 22;
  in.struct_size = sizeof(uint64_t);                                                          /* too short to hold even `img` */
  if (forward(h, 1, &in, &out, 0) != 1 || !strstr(last_error(), "poco_inputs_t.struct_size")) return 23;
  if (status(h) != 0) return 26;                                                              /* nothing ran, nothing timed out */
  destroy(h);
  {
    /* a tile configuration is SEVEN ints {MT,NT,WM,WN,R,NI,ALG} (the header said six until round 5: a client written from it
     * under-allocated by one int per configuration); the argument check runs before anything touches a GPU */
    int cfgs7[2 * 7] = {4, 2, 2, 2, 8, 1, 1, 4, 2, 2, 2, 8, 1, 0};
    float ms[2] = {0.f, 0.f};
    if (tune(1, 8, 8, 15, 16, 3, 1, cfgs7, 2, 1, ms, 0) != 1 || !strstr(last_error(), "poco_tune_conv")) return 24;   /* Cin % 16 != 0 */
    if (tune(1, 8, 8, 16, 16, 3, 1, cfgs7, 0, 1, ms, 0) != 1) return 25;                                               /* ncfg < 1 */
    if (tune(1, 8, 8, 16, 32, 3, 1, cfgs7, 2, -1, ms, 0) != 1 || !strstr(last_error(), "residual")) return 27;         /* residual form needs Cin == Cout */
  }
  if (create_ex("resnet50-cliff", 4, 1, "no_such_option=1", &h) == 0) return 16;     /* unknown build option is an error ... */
  if (strstr(last_error(), "no_such_option") == 0) return 17;                         /* ... that names it */
  if (create_ex("resnet50-cliff", 4, 1, "dual=0,flow_ctx_rows=96,debug_wait_spins=1000,debug_mlp_timeouts=1", &h) != 0) return 18;
  if (num_tensors(h) != n) return 19;                                                 /* the separate-launch form consumes the same tensors */
  destroy(h);
  printf("ok %d\n", n);
  return 0;
}
