// PnP / BA device back-end (filled in below).
#include "pmv_ctx.h"
namespace pmv {
struct BackendBuffers { int dummy; };
int backend_create(pmv_ctx* c) { c->be = new BackendBuffers(); return PMV_OK; }
void backend_destroy(pmv_ctx* c) { delete c->be; c->be = nullptr; }
}
