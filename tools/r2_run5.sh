#!/bin/bash
for d in 0 1 2 3; do echo "dry=$d"; IVX_SOLVER_DRY=$d python tools/time_pile.py 16 --groups 5 2>&1 | tail -1 | cut -c1-230; done
