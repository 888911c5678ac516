"""Shared set-up of the headless drivers: put the drop-in modules in front of everything else."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "aircraftoptimalcontrol_amd", "dropin"))
