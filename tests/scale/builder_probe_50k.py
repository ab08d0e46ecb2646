"""Anchor of the builder oracle to the only numbers the REAL reference builder ever produced in this project: the survey's
probe of the compiled reference (N = 50 000 x 200-d OOD inner-product set of the 8(d) recipe,
canonical build parameters -> average degree 33.1, longest row 70, recall@10 0.915 / 0.997 at L_pq = 20 / 100, hops 23.7 /
102.5, distance evaluations 1 066 / 3 519).  Runs the restated builder (oracle/roar_builder_oracle.c) on the same recipe and
writes tests/golden/builder_probe_50000.json with both sets of figures; tests/test_oracle_pinning.py holds the restated
builder to the survey's figures, tests/test_gpu_parity.py holds roar_build_roargraph to both.  A few minutes of one host core."""
import os, subprocess, sys

here = os.path.dirname(os.path.abspath(__file__))
sys.exit(subprocess.call([sys.executable, os.path.join(here, "builder_yardstick.py"), "--n", "50000", "--name", "builder_probe",
                          "--survey-probe", "--Ls", "10", "20", "50", "100", "200", "500"]))
