"""Parameter tables and MJCF builders of the locomotion models."""
