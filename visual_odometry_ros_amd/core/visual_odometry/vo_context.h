// vo_context.h — RAII owner of a vo_ctx shared by the host-side classes.
#ifndef VO_AMD_CONTEXT_H_
#define VO_AMD_CONTEXT_H_

#include <memory>
#include <stdexcept>
#include <string>

#include "../../../include/vo_hip.h"

namespace vo {

class Context {
 public:
  explicit Context(int device = 0, int max_width = 1241, int max_height = 376, int max_points = 4096,
                   int n_slots = 4, int max_level = 6) {
    vo_config cfg{device, max_width, max_height, max_points, n_slots, max_level};
    const int rc = vo_create(&cfg, &ctx_);
    if (rc != VO_OK) {
      std::string msg = vo_last_error(ctx_);
      if (ctx_) vo_destroy(ctx_);
      ctx_ = nullptr;
      throw std::runtime_error("libvo_hip: " + msg);  // no CPU fallback
    }
    n_slots_ = n_slots;
  }
  ~Context() {
    if (ctx_) vo_destroy(ctx_);
  }
  Context(const Context &) = delete;
  Context &operator=(const Context &) = delete;
  vo_ctx *get() const { return ctx_; }
  int n_slots() const { return n_slots_; }
  // maps the reference's throw sites / return-false onto the C status codes
  int check(int rc) const {
    if (rc < 0) throw std::runtime_error(vo_last_error(ctx_));
    return rc;
  }

 private:
  vo_ctx *ctx_ = nullptr;
  int n_slots_ = 0;
};
using ContextPtr = std::shared_ptr<Context>;

}  // namespace vo
#endif
