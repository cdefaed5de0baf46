/*
 * rrt_tile_objects.h -- the host objects behind rrt_tile_map (explicit tile -> shard assignment) and rrt_tile_order (cost-ordered
 * dispatch) and their registries.  Included by rrt_hip.hip inside its anonymous namespace, after rrt_kernels.h (RowMap) (round 6:
 * split out of rrt_hip.hip, nothing else changed).  The sort behind rrt_tile_order: rrt_tile_sort.h.
 */
#ifndef RRT_TILE_OBJECTS_H
#define RRT_TILE_OBJECTS_H

/* ------------------------------------------------------------------ explicit tile -> shard maps (rrt_tile_map)
 * SURVEY.md 8e's "cost-model-weighted assignment": instead of tile t -> shard t mod G, any assignment.  The object keeps a
 * device image of the lists the kernels index (a shard's tiles in increasing t; every tile's shard and place), tied to
 * the device it was created on. */
struct TileMapObject {
    int height, tile_rows, n_shards, n_tiles, device;
    std::vector<int> shard_of_tile, offset, rows;    /* offset[s]: where shard s's tiles start in tile_of_local */
    int* d_img = nullptr;
};
std::mutex g_tm_mu;
std::unordered_map<int, std::shared_ptr<TileMapObject>> g_tm;
int g_tm_next = 1;
std::shared_ptr<TileMapObject> tile_map_lookup(int id) {
    std::lock_guard<std::mutex> lk(g_tm_mu);
    auto it = g_tm.find(id);
    return it == g_tm.end() ? nullptr : it->second;
}

/* ------------------------------------------------------------------ cost-ordered dispatch (rrt_tile_order)
 * Workgroups are dispatched in blockIdx order and a wave tile's cost is only known once it has been rendered, so the
 * static order (row blocks from the middle outwards) is right for the reference's default view and wrong wherever the
 * longest rays are somewhere else: from inside the disk a 4K frame spends 8 % of its time draining.
 * An rrt_tile_order object remembers, per wave tile, the clocks the previous launch through it took, and dispatches the
 * next launch of the same geometry longest-first (one radix sort of n_tiles keys after the frame, ~20 us: rrt_tile_sort.h).  Any order
 * renders the same pixels.  All launches through one object are serialised on the device (an event chains them across
 * streams): frames that should overlap need an object each. */
struct TileOrderObject {
    int device;
    unsigned* d_cost;        /* clocks >> 4 of each wave tile, written by the render kernel */
    unsigned* d_sorted;      /* the sort's keys between its two passes (before the sort: scratch of the cost probe) */
    unsigned* d_iota;        /* ... and the tiles that travel with them */
    unsigned* d_perm[2];     /* dispatch slot -> wave tile; [cur] is the one the next matching launch reads */
    void* d_temp; size_t temp_bytes;      /* the sort's (digit, block) counters (rrt_tile_sort.h) */
    size_t n_cap;
    int cur;
    bool have;               /* d_perm[cur] holds an order for the geometry below */
    unsigned grid_x, grid_y;
    int width, height; RowMap rows;
    hipEvent_t chained;      /* after the last sort */
    unsigned long long launches, ordered, seeded;
    bool no_seed;            /* rrt_tile_order_set_seeding(id, 0): a geometry without history renders in the static order */
    bool dead;               /* destroyed (a launch that was waiting for `mu` must not touch the buffers) */
    std::mutex mu;           /* launches through one object are serialised on the host as well */
};
bool same_row_map(const RowMap& a, const RowMap& b) {
    return a.n_local_rows == b.n_local_rows && a.y_base == b.y_base && a.tile_rows == b.tile_rows && a.shard == b.shard &&
           a.n_shards == b.n_shards && a.tile_of_local == b.tile_of_local;
}
/* the registry lock only covers the lookup; an object is pinned by its shared_ptr and serialised by its own mutex, so
 * threads driving different objects (different GPUs) never wait for each other */
std::mutex g_to_mu;
std::unordered_map<int, std::shared_ptr<TileOrderObject>> g_to;
int g_to_next = 1;
std::shared_ptr<TileOrderObject> tile_order_lookup(int id) {
    std::lock_guard<std::mutex> lk(g_to_mu);
    auto it = g_to.find(id);
    return it == g_to.end() ? nullptr : it->second;
}

#endif /* RRT_TILE_OBJECTS_H */
