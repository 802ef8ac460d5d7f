"""Prints the RCCL libraries mapped into a process that has imported torch and made a communicator id through
librtg.so: one path means the run-time binding found the copy the process already carries."""
import os
import sys

import torch  # noqa: F401
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import rtg
uid = rtg.comm_unique_id()
print(sorted({l.split()[-1] for l in open("/proc/self/maps") if "rccl" in l}))
