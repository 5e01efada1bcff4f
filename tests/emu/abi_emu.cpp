// Identity of the CPU lane-emulator build of the kernels (tests only).
extern "C" const char* tzr_backend(void) { return "emu"; }
extern "C" int tzr_abi_version(void) { return 15; }

// The native step driver (csrc/step_driver.hip: host-only code) is compiled into the emulator UNCHANGED: hipGraphLaunch / events /
// streams are the synchronous shims of tests/emu/hip/hip_runtime.h, RCCL is whatever library `tzr_comm_create` is pointed at
// (the CPU suite: tests/emu/rccl_stub.cpp, shared memory between the ranks' processes).  A "captured graph" on this side is a
// host function recorded by the test harness:
#include <hip/hip_runtime.h>
extern "C" int tzr_emu_graph_create(void (*fn)(void*), void* arg, void** out_graph_exec) {
  if (!fn || !out_graph_exec) return -1;
  *out_graph_exec = new hipGraphExecEmu{fn, arg};
  return 0;
}
extern "C" int tzr_emu_graph_destroy(void* graph_exec) {
  delete static_cast<hipGraphExecEmu*>(graph_exec);
  return 0;
}

// Context switch of the lane fibers (tests/emu/hip/hip_runtime.h): saves the callee-saved registers of
// the System V x86-64 ABI on the current stack, stores that stack pointer, loads the other one.
__asm__(R"(
.text
.globl tzr_emu_switch
.type tzr_emu_switch, @function
tzr_emu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size tzr_emu_switch, .-tzr_emu_switch
)");
