// Training path (train-mode BatchNorm, backward, optimizer) — entry points of
// include/sykepic_hip.h that stand in for sykepic/train/train.py:239-243.
#include "model.h"

struct TrainState {};

void spk_train_free(spk_model* m) {
  delete m->train;
  m->train = nullptr;
}

extern "C" int spk_train_forward_backward(spk_model* m, const void* x, int n, int h, int w, int layout,
                                          int dtype, const int64_t* y, float* stats, float* logits) {
  spk_set_error("spk_train_forward_backward: not built yet");
  return SPK_ERR_UNSUPPORTED;
}
extern "C" int spk_optim_step(spk_model* m, const spk_optim_desc* opt) {
  spk_set_error("spk_optim_step: not built yet");
  return SPK_ERR_UNSUPPORTED;
}
extern "C" int spk_model_grad_buffer(spk_model* m, void** dev_ptr, int64_t* numel) {
  spk_set_error("spk_model_grad_buffer: not built yet");
  return SPK_ERR_UNSUPPORTED;
}
extern "C" int spk_model_read_grad(spk_model* m, const char* key, void* host, int64_t numel) {
  spk_set_error("spk_model_read_grad: not built yet");
  return SPK_ERR_UNSUPPORTED;
}
