// mifc_env.hip -- see mifc_env.h (host code only).
#include "mifc_env.h"

#include <cstdlib>
#include <cstring>
#include <mutex>

namespace mifc {

namespace {
Env g_env;
std::mutex g_env_mutex;

int positive_int(const char* name)
{
  const char* e = std::getenv(name);
  if (!e)
    return 0;
  const int v = std::atoi(e);
  return v > 0 ? v : 0;
}

bool not_zero(const char* name)
{
  const char* e = std::getenv(name);
  return !(e && e[0] == '0');
}
} // namespace

const Env& env()
{
  return g_env;
}

void env_reload()
{
  Env e;
  e.force_cell_kernel = std::getenv("MIFC_FORCE_CELL_KERNEL") != nullptr;
  e.host_pipeline = not_zero("MIFC_HOST_PIPELINE");
  e.fused2 = not_zero("MIFC_FUSED2");
  e.shapiro_fused = not_zero("MIFC_SHAPIRO_FUSED");
  e.shapiro_regs = not_zero("MIFC_SHAPIRO_REGS");
  e.ewise_max_blocks = positive_int("MIFC_EWISE_MAX_BLOCKS");
  if (const char* s = std::getenv("MIFC_SCALAR_ROWS_R")) {
    const int v = std::atoi(s);
    e.scalar_rows_r = v > 0 ? v : 0;
  }
  e.fused2_band = positive_int("MIFC_FUSED2_BAND");
  e.host_threads = positive_int("MIFC_HOST_THREADS");
  e.host_chunk_mib = positive_int("MIFC_HOST_CHUNK_MIB");
  e.derived_blocks = positive_int("MIFC_DERIVED_BLOCKS");
  e.derived_pipe = not_zero("MIFC_DERIVED_PIPE") ? 1 : 0;
  e.levelwalk = not_zero("MIFC_VORTDIV_LEVELWALK");
  e.split_roles = not_zero("MIFC_VORTDIV_SPLIT");
  e.levelwalk_min_units = positive_int("MIFC_LEVELWALK_MIN_UNITS");
  e.slab_graph = not_zero("MIFC_SLAB_GRAPH");
  e.ragged_split = not_zero("MIFC_RAGGED_SPLIT");
  if (const char* s = std::getenv("MIFC_VORTDIV_TUNE")) {
    e.has_vortdiv_tune = true;
    std::strncpy(e.vortdiv_tune, s, sizeof e.vortdiv_tune - 1);
  }
  if (const char* s = std::getenv("MIFC_SCALAR_SPLIT_TUNE"))
    std::strncpy(e.scalar_split_tune, s, sizeof e.scalar_split_tune - 1);
  std::lock_guard<std::mutex> lock(g_env_mutex);
  g_env = e;
}

} // namespace mifc
