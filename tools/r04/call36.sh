#!/bin/bash
bash tools/regen_profiles.sh r04 2bab729
