// shard_plan.h -- who runs what when ONE track is spread over the GPUs of a node (BASELINE config 4).
//
// Two independent axes of the reference's work:
//   * source model: the per-target loop of umx_inference (inference.cpp:70-186) shares nothing between targets until
//     wiener_filter (inference.cpp:192-193) -- G target groups;
//   * segment: split_inference's chunks (umx.cpp:214-227) are coupled only through the per-layer LSTM state of the same
//     target (layer l of segment s needs layer l's (h, c) of segment s-1) -- a pipeline of P stages.
// world = G x P.  Rank r is (group g = r % G, stage p = r / G): it runs the targets t with t % G == g on the segments s
// with s % P == p.  Of the G ranks that share a segment, one -- rotating with the segment index so that the work is
// spread -- receives the other groups' target magnitudes, runs the Wiener filter + inverse STFT and owns the stems.
// Shared by host/split.cpp (host buffers, any transport; the CPU tests drive it over gloo) and host/mgpu.cpp (device
// buffers, RCCL).
#pragma once

namespace umx_plan
{
struct Plan
{
    int world, G, P;
    int group(int rank) const { return rank % G; }
    int stage(int rank) const { return rank / G; }
    int rank_of(int g, int p) const { return g + G * p; }
    bool owns_target(int rank, int t) const { return t % G == group(rank); }
    int owner_of_target(int t, int p) const { return rank_of(t % G, p); } // on stage p
    int stage_of_segment(int s) const { return s % P; }
    bool runs_segment(int rank, int s) const { return stage_of_segment(s) == stage(rank); }
    int wiener_rank(int s) const { return rank_of((s / P) % G, s % P); }
    // the rank that ran segment s - 1 / will run s + 1 for the same targets
    int prev_rank(int rank, int s) const { return rank_of(group(rank), (s - 1) % P); }
    int next_rank(int rank, int s) const { return rank_of(group(rank), (s + 1) % P); }
};

inline int gcd4(int world) { return world % 4 == 0 ? 4 : world % 2 == 0 ? 2 : 1; }

// by_target == false: segments only (G = 1: the "carry" mode); true: as many target groups as divide both 4 and world
inline Plan make_plan(int world, bool by_target)
{
    Plan pl;
    pl.world = world;
    pl.G = by_target ? gcd4(world) : 1;
    pl.P = world / pl.G;
    return pl;
}

// Colour of the state-ring edge that leaves stage p (towards stage (p + 1) % P): adjacent edges differ, so a rank never
// uses one communicator for its outgoing AND its incoming edge (it sends on a stream of its own; RCCL serialises the
// operations of one communicator, which would tie the send to the receive and, with rendezvous sends, close a cycle).
// Even rings need two colours, odd rings three.
inline int edge_colour(int p, int P) { return (P % 2 == 1 && p == P - 1 && P > 1) ? 2 : (p & 1); }
} // namespace umx_plan
