"""Mirror of /root/reference/modules/pna (PNA node embedding) on the HIP multi-aggregator kernel."""
