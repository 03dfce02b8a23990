"""Helpers of bench.py (the entry point the driver runs stays bench.py): workloads, CPU baseline, end-to-end legs, HBM traffic."""
