"""Probe (GPU box): may a process that has initialised the GPU start a child program (fork + exec with close_fds)? The 2-rank
bench test (tests/test_gpu_round4.py) depends on it."""
import subprocess, sys, torch
torch.cuda.init(); x = torch.ones(4, device="cuda"); torch.cuda.synchronize()
r = subprocess.run([sys.executable, "-c", "print('child ok')"], capture_output=True, text=True, timeout=120)
print("rc", r.returncode, "out", r.stdout.strip(), "err", r.stderr.strip()[-300:])
