; ppcx_step_kernel, optimised LLVM IR out of clang -O3 (before the AMDGPU backend), sources at 02dec93 minus the asm barrier.
; s_st + 76 = ChainScalars::eps_attempt, s_st + 80 = eps_call; Cmd::rng_c1 = eps_call, Cmd::rng_c3 = eps_attempt (ppcx_nuts.h issue_eps_try).
; The (rng_c1, rng_c3) pair is ONE <2 x i32> phi made by the SLP vectoriser; block %1540 is issue_eps_try reached from BOTH the
; first trial (pred %1522: eps_dir == 0) and the halving / doubling path (pred %1533). Its incoming value %515 is correct:

747:  %514 = load <2 x i32>, ptr addrspace(3) getelementptr inbounds nuw (i8, ptr addrspace(3) @_ZZN4ppcx16ppcx_step_kernelENS_8StepArgsEE4s_st, i32 76), align 4, !tbaa !15
748:  %515 = shufflevector <2 x i32> %514, <2 x i32> poison, <2 x i32> <i32 1, i32 0>

1540:                                             ; preds = %1533, %1522
  %1541 = phi i32 [ %1524, %1522 ], [ %513, %1533 ]
  %1542 = phi double [ %519, %1522 ], [ %1536, %1533 ]
  %1543 = add nsw i32 %517, 1
  %1544 = insertelement <2 x i32> poison, i32 %1541, i64 0
  %1545 = insertelement <2 x i32> %1544, i32 %1543, i64 1
  br label %4131

4790:  %4153 = phi <2 x i32> [ zeroinitializer, %4130 ], [ %1509, %1496 ], [ %515, %1540 ], [ %1851, %1546 ], [ %1474, %1467 ], [ zeroinitializer, %1458 ], [ %1490, %1479 ], [ zeroinitializer, %1492 ], [ zeroinitializer, %2553 ], [ %2811, %2592 ], [ %2590, %2575 ], [ zeroinitializer, %2874 ], [ zeroinitializer, %2860 ], [ zeroinitializer, %3616 ], [ %3875, %3656 ], [ %3654, %3639 ], [ zeroinitializer, %4023 ], [ zero
  ...
6611:  %5781 = extractelement <2 x i32> %4153, i64 0
6612:  store i32 %5781, ptr addrspace(3) @_ZZN4ppcx16ppcx_step_kernelENS_8StepArgsEE4s_nc.11, align 8, !tbaa !15
6613:  %5782 = extractelement <2 x i32> %4153, i64 1
6614:  store i32 %5782, ptr addrspace(3) @_ZZN4ppcx16ppcx_step_kernelENS_8StepArgsEE4s_nc.12, align 4, !tbaa !15
