"""CPU oracle for the embedding + matching path — test infrastructure only (see face_oracle.py)."""
