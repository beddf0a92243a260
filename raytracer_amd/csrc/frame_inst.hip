// frame_inst.hip -- one instantiation of the frame's launch chain and of every kernel it launches.  Compiled once per
// (RT355_REAL, RT355_MODE) pair: double / float x the kernel modes 0..3 (M_COUNT | M_INST), each into an object of its
// own, so the pairs build side by side.
#include "frame.h"

#if !defined(RT355_REAL) || !defined(RT355_MODE)
#error "compile with -DRT355_REAL=double|float -DRT355_MODE=0..3"
#endif

template int rt355::run_generations<RT355_REAL, RT355_MODE>(rt355_ctx*, const rt355_scene*, rt355::RayPool<RT355_REAL>&, rt355::FrameCtl*, uint32_t, int,
                                                             const rt355::DevCamera<RT355_REAL>&, bool, int, float*, LaunchLog&);
