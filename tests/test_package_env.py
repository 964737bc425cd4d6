"""nnr_amd/__init__.py pins HIP's hardware-queue count (GPU_MAX_HW_QUEUES) to the value the step is tuned on, and says so when the caller chose another
(round 6: 5 queues and more measured 1.4-2.1x slower steps; profiles/r06_ab.txt calls 42-43).  No GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(code, **env):
    e = {k: v for k, v in os.environ.items() if k != 'GPU_MAX_HW_QUEUES'}
    e.update(env)
    return subprocess.run([sys.executable, '-W', 'always', '-c', code], capture_output=True, text=True, cwd=ROOT, env=e)


def test_hardware_queue_count_is_pinned_when_the_caller_has_not_chosen():
    r = _run('import os, nnr_amd; print(os.environ["GPU_MAX_HW_QUEUES"], os.environ["HIP_FORCE_DEV_KERNARG"])')
    assert r.returncode == 0 and r.stdout.split() == ['4', '1'] and 'GPU_MAX_HW_QUEUES' not in r.stderr


def test_another_hardware_queue_count_is_kept_and_warned_about():
    r = _run('import os, nnr_amd; print(os.environ["GPU_MAX_HW_QUEUES"])', GPU_MAX_HW_QUEUES='8')
    assert r.returncode == 0 and r.stdout.strip() == '8' and 'GPU_MAX_HW_QUEUES=8' in r.stderr and 'SLOWER' in r.stderr
    r = _run('import nnr_amd', GPU_MAX_HW_QUEUES='4')
    assert r.returncode == 0 and r.stderr.strip() == ''
