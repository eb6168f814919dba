#!/bin/sh
# TEST INFRASTRUCTURE: builds the CPU library behind tests/tile_model.py (host tables of the tiles layout; no GPU code)
set -e
cd "$(dirname "$0")"
mkdir -p _build
g++ -O2 -std=c++17 -shared -fPIC -Wall -o _build/libtiletables.so tile_tables.cpp
