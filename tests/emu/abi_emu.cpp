// Identity of the CPU lane-emulator build of the kernels (tests only).
extern "C" const char* tzr_backend(void) { return "emu"; }
extern "C" int tzr_abi_version(void) { return 9; }

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
