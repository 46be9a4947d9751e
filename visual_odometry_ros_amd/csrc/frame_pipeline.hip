#include "vo_internal.hpp"
#include "vo_kernels.hpp"
void vo_frame_free(vo_ctx *c) { (void)c; }
