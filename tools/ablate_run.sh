#!/bin/bash
python tools/ablate.py snapshot
python tools/ablate.py time
for f in safe_adaptation_gym_amd/libsag_*.so; do SAG_LIB=$PWD/$f python tools/ablate.py time; done
