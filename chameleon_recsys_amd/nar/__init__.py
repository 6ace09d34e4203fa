"""Host-side mirror of the reference's ``nar_module/nar`` package for the NAR training path."""
