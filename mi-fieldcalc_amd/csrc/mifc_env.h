// mifc_env.h -- the tuning / diagnostic environment variables of the library,
// read ONCE (mifc_create, mifc_reload_env) into a process-wide snapshot.  No
// launch path calls getenv: the launchers read this struct.  The variables
// select between code paths that all give the reference's results (A/B
// measurements, tests of every fallback); none of them changes what is computed.
#ifndef MIFC_ENV_H
#define MIFC_ENV_H

namespace mifc {

struct Env
{
  bool force_cell_kernel = false; // MIFC_FORCE_CELL_KERNEL: one-lane-per-cell stencil kernels everywhere
  bool host_pipeline = true;      // MIFC_HOST_PIPELINE=0: stage host-resident level batches whole
  bool fused2 = true;             // MIFC_FUSED2=0: multi-pass thermalFrontParameter / plevelqvector
  bool shapiro_fused = true;      // MIFC_SHAPIRO_FUSED=0: four-launch shapiro2_filter
  bool shapiro_regs = true;       // MIFC_SHAPIRO_REGS=0: the one-launch form with its rows in LDS rings instead of registers
  int ewise_max_blocks = 0;       // MIFC_EWISE_MAX_BLOCKS (> 0 overrides the grid cap of the table kernels)
  int scalar_rows_r = -1;         // MIFC_SCALAR_ROWS_R: -1 unset, 0 = "set, keep the default height", > 0 band height
  int fused2_band = 0;            // MIFC_FUSED2_BAND (> 0 overrides the band height of the fused two-stage kernels)
  int host_threads = 0;           // MIFC_HOST_THREADS (> 0)
  int host_chunk_mib = 0;         // MIFC_HOST_CHUNK_MIB (> 0)
  int derived_blocks = 0;         // MIFC_DERIVED_BLOCKS (> 0 overrides the persistent grid of the fused derived kernel)
  int derived_pipe = 1;           // MIFC_DERIVED_PIPE=0: the fused derived kernel without the two-trip software pipeline (A/B)
  bool levelwalk = true;          // MIFC_VORTDIV_LEVELWALK=0: deep wind batches on the row-walking kernel (A/B)
  bool split_roles = true;        // MIFC_VORTDIV_SPLIT=0: deep batches of the stencil operators on the level-walking kernels whose waves load AND store (A/B)
  int levelwalk_min_units = 0;    // MIFC_LEVELWALK_MIN_UNITS (> 0 overrides the launch size from which the level-walking forms are chosen; tests)
  bool ragged_split = true;       // MIFC_RAGGED_SPLIT=0: widths that are not a multiple of 4 keep the flat four-cells-per-lane kernel for deep batches too (A/B)
  bool slab_graph = true;         // MIFC_SLAB_GRAPH=0: the row-slab step (mifc_slab_plan_step) enqueues its sequence directly instead of replaying a HIP graph
  bool has_vortdiv_tune = false;  // MIFC_VORTDIV_TUNE="R=8,D=1,..."
  char vortdiv_tune[256] = {0};
  char scalar_split_tune[64] = {0}; // MIFC_SCALAR_SPLIT_TUNE="TR=12,NL=2,PF=2,LG=6": shape of the split-role form of the one-input stencil operators (A/B, tests)
};

// the current snapshot (defaults until the first reload)
const Env& env();
// re-reads the process environment into the snapshot
void env_reload();

} // namespace mifc

#endif // MIFC_ENV_H
