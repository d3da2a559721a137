"""polus.mock: the world-size-1 stand-in for horovod (polus/mock/horovod.py:5-24)."""
