"""placeholder: imported by demo.py:25, never called on the sampling path"""
