#!/bin/bash
bash tools/regen_profiles.sh r04 5cd1181
