// TEST INFRASTRUCTURE (see ../cuda_runtime_api.h).  kfusion/kinfu.hpp -> warp_field_optimiser.hpp names the Opt solver's classes in
// two declarations; Opt / terra / mLib are third-party and absent.  Nothing in the files compiled here uses them.
#pragma once
class CombinedSolver;
struct CombinedSolverParameters {};
