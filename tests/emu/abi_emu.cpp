// Identity of the CPU lane-emulator build of the kernels (tests only).
extern "C" const char* tzr_backend(void) { return "emu"; }
extern "C" int tzr_abi_version(void) { return 11; }

// The native step driver (csrc/step_driver.hip: hipGraphLaunch + RCCL, host-only code) has nothing to emulate: the
// emulator library exports its entry points so that the binding table loads, and refuses them.
#include <cstddef>
#include <cstdint>
#define TZR_EMU_UNSUPPORTED (-4)
extern "C" int tzr_comm_available(const char*) { return 0; }
extern "C" int tzr_comm_version(const char*) { return -1; }
extern "C" int tzr_comm_unique_id(const char*, void*, size_t) { return TZR_EMU_UNSUPPORTED; }
extern "C" int tzr_comm_create(const char*, const void*, size_t, int, int, void**) { return TZR_EMU_UNSUPPORTED; }
extern "C" int tzr_comm_destroy(void*) { return 0; }
extern "C" int tzr_comm_all_to_all(void*, const void*, void*, int64_t, void*) { return TZR_EMU_UNSUPPORTED; }
extern "C" int tzr_comm_all_reduce(void*, float*, int64_t, int, void*) { return TZR_EMU_UNSUPPORTED; }
extern "C" int tzr_step_create(void**) { return TZR_EMU_UNSUPPORTED; }
extern "C" int tzr_step_destroy(void*) { return 0; }
extern "C" int tzr_step_add_graph(void*, void*) { return TZR_EMU_UNSUPPORTED; }
extern "C" int tzr_step_add_all_to_all(void*, void*, const void*, void*, int64_t, int) { return TZR_EMU_UNSUPPORTED; }
extern "C" int tzr_step_add_all_reduce(void*, void*, float*, int64_t, int, int) { return TZR_EMU_UNSUPPORTED; }
extern "C" int tzr_step_add_wait(void*, int) { return TZR_EMU_UNSUPPORTED; }
extern "C" int tzr_step_num_ops(void*) { return TZR_EMU_UNSUPPORTED; }
extern "C" int tzr_step_run(void*, void*) { return TZR_EMU_UNSUPPORTED; }

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
