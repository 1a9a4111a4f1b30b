/* Link-time stand-ins for librt_mi355x (our own C-ABI) so that the compiled host can be linked and run under the CPU sanitizers:
   the run exercises the scene loader and the marshal only; every device call reports failure. */
#include <stddef.h>
struct rt_ctx; 
int rt_set_params(struct rt_ctx* c, const void* p) { return -1; }
int rt_upload_spheres(struct rt_ctx* c, const void* p, int n) { return -1; }
int rt_upload_triangles(struct rt_ctx* c, const void* p, int n) { return -1; }
int rt_upload_meshinfo(struct rt_ctx* c, const void* p, int n) { return -1; }
int rt_render(struct rt_ctx* c, int a, int b) { return -1; }
int rt_reset_accum(struct rt_ctx* c) { return -1; }
int rt_read_accum(struct rt_ctx* c, float* f, size_t n) { return -1; }
int rt_upload_local_meshes(struct rt_ctx* c, const void* t, int nt, const void* ch, int nc, int nm) { return -1; }
int rt_set_mesh_transforms(struct rt_ctx* c, const void* x, int n) { return -1; }
const char* rt_last_error(struct rt_ctx* c) { return "stub"; }
struct rt_multi;
int rt_multi_set_params(struct rt_multi* m, const void* p) { return -1; }
int rt_multi_upload_spheres(struct rt_multi* m, const void* p, int n) { return -1; }
int rt_multi_upload_triangles(struct rt_multi* m, const void* p, int n) { return -1; }
int rt_multi_upload_meshinfo(struct rt_multi* m, const void* p, int n) { return -1; }
int rt_multi_render(struct rt_multi* m, int a, int b) { return -1; }
int rt_multi_reset_accum(struct rt_multi* m) { return -1; }
int rt_multi_read_accum(struct rt_multi* m, float* f, size_t n) { return -1; }
int rt_multi_upload_local_meshes(struct rt_multi* m, const void* t, int nt, const void* ch, int nc, int nm) { return -1; }
int rt_multi_set_mesh_transforms(struct rt_multi* m, const void* x, int n) { return -1; }
const char* rt_multi_last_error(const struct rt_multi* m) { return "stub"; }
