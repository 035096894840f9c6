#!/bin/bash
for s in 0 10 20 30 40 50 60 80; do VQHIP_SCREEN_STAGGER=$s python tools/time_chain_stage.py 256 1024 18; done
for s in 0 20 40 60; do VQHIP_CHAIN_NOWRITE=1 VQHIP_SCREEN_STAGGER=$s python tools/time_chain_stage.py 256 1024 18; done
for s in 0 20 40 80; do VQHIP_SCREEN_STAGGER=$s python tools/time_chain_stage.py 128 4096 18; done
VQHIP_CHAIN_NOWRITE=1 python tools/time_chain_stage.py 128 4096 18
