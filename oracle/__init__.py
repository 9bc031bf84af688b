"""oracle/ — TEST INFRASTRUCTURE (CPU restatement of the reference hot path).  See reference_math.py."""
