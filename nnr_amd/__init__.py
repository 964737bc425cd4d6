"""nnr_amd: the CNE + SUE (and MHSA + MHSA) training step of Veason-silverbullet/NNR on MI355X -- host side of include/nnr_hip.h."""
import os as _os
import warnings as _warnings

# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The step runs on 5-6 HIP streams and is tuned on 4 queues;
# with 5 or more the CU-pair recurrence and everything behind it falls off a cliff (batch 64: 9.5 -> 13 ms per step, batch 8: 3.1 -> 6.5; 2-3 queues
# serialise the streams: profiles/r06_ab.txt calls 42-43).  Pinned to HIP's own default when the caller has not chosen -- effective when this package
# is imported before the process's first HIP call, as bench.py / the trainer do -- and a loud warning when the caller has chosen another value.
_q = _os.environ.get('GPU_MAX_HW_QUEUES')
if _q is None:
    _os.environ['GPU_MAX_HW_QUEUES'] = '4'
elif _q.strip() != '4':
    _warnings.warn('nnr_amd: GPU_MAX_HW_QUEUES=%s -- the training step is tuned on HIP\'s default of 4 hardware queues; 5 and more measured 1.4-2.1x '
                   'SLOWER steps, 2-3 serialise its streams (profiles/r06_ab.txt calls 42-43)' % _q)
del _q
# Kernel arguments in device memory (HIP_FORCE_DEV_KERNARG=1, this ROCm's default): with 0 the step's ~230 dependent launches cost +0.15 ms at batch 64, +0.13 ms at batch 8,
# +0.08 ms on the MHSA step (profiles/r06_ab.txt call 57).  Made explicit when the caller has not chosen.
_os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
