// The body of one observation step: textually included by update_kernel (one step per launch; the kernel's by-value parameters
// are used in place: taking references to them would make the compiler keep private copies of the argument structs) and by
// persist_step (one iteration of the persistent loop).  Names the including function provides: METHOD, MODE, BIG, PERSIST,
// PP (constants); h_tot, h_ctrl, h_front, h_nb, h_slot, h_grid, h_flags, B, PA, U, rng, ws, ppa (update_kernel's parameters, or what stands
// in for them in an iteration); the macros STEP_WPAR (ws.wpar), STEP_PLAN (U.plan_inline != 0) and STEP_FRONT (U.front), which a per-step
// launch reads in place (a local copy of each costs the distributed variants a scalar register they do not have); qe (LDS copy of the iteration's queue
// entry, PERSIST only); carry (StepCarry, PERSIST only).
    constexpr bool ARR = MODE == 1, DIST = MODE == 2;
    static_assert(!PERSIST || (MODE == 0 && !BIG), "the persistent loop: compact single contexts");
    const int bid = PERSIST ? (int) blockIdx.x / kPersistStride : (int) blockIdx.x;  // this workgroup's number among those that work
    __shared__ float sh_w[kBlock / kWave], sh_w2[kBlock / kWave];
    // landmarks re-observed this step, staged between the proposal pass and the likelihood/feature-update pass
    // (each thread only touches its own column: no barrier, no bank conflict: consecutive lanes, consecutive slots)
    // dynamic LDS (sized by the launcher, staging_slots()): [slots][256] float4 + [slots][256] float of staged records,
    // then -- inline plan only -- [nblocks + 1] doubles: exclusive prefix of the previous step's block totals.  Sizing the
    // staging by the packet (0 / 4 / 8 landmarks) instead of a static 40 KB keeps 5-8 blocks per CU resident at the
    // webmap's 3.5 landmarks per step instead of 3.
    // Layout (every offset a function of the preloaded head arguments only, so that the scan can start before the argument
    // structs have arrived): [prefix of the block totals][ancestor windows][staged records A][staged records B]
    extern __shared__ __align__(16) unsigned char dyn_lds[];
    const int nbg = DIST ? h_nb * B.n_shards : h_nb;  // blocks of the whole particle set
    const bool h_plan = PERSIST ? STEP_PLAN : (h_flags & 1) != 0, h_scan_global = !PERSIST && (h_flags & 2) != 0;
    const bool lay_plan = PERSIST || h_plan;  // (the persistent launch is sized for planning iterations)
    double *const off = reinterpret_cast<double *>(dyn_lds);
    const size_t off_bytes = (lay_plan && !h_scan_global) ? sizeof(double) * (((size_t) nbg + 3) & ~(size_t) 1) : 0;
    // ... then, launches that plan inline: the per-wave windows of the ancestor search (find_ancestor_win)
    float *const wins = reinterpret_cast<float *>(dyn_lds + off_bytes);  // 16-byte aligned
    const size_t win_bytes = lay_plan ? update_window_bytes() : 0;
    float *const win = wins + (threadIdx.x / kWave) * (kWinBlocks * kBlock);  // this wave's window
    float4 *const shA = reinterpret_cast<float4 *>(dyn_lds + off_bytes + win_bytes);
    const int nslots = staging_slots(METHOD, BIG, U.m);
    float *const shB = reinterpret_cast<float *>(dyn_lds + off_bytes + win_bytes + (size_t) nslots * kBlock * sizeof(float4));
    // per-particle association (PP): behind the staged records, the particle's observation index of every staged entry and -- when
    // they fit -- the step's observations themselves (a load from global memory inside the passes' bodies makes the wave wait for
    // every record prefetch in flight as well: the note on the packet below)
    [[maybe_unused]] int32_t *const shJ = PP ? reinterpret_cast<int32_t *>(shB + (size_t) nslots * kBlock) : nullptr;
    [[maybe_unused]] float *const shZ = PP ? reinterpret_cast<float *>(shJ + (size_t) nslots * kBlock) : nullptr;
    __shared__ double sh_a[kBlock / kWave], sh_q[kBlock / kWave];
    __shared__ EstItem sh_est[kBlock / kWave];
    SLAM_STAMP(0);  // kernel entry
    const size_t S = (size_t) B.ncap;
    Ctrl *ctrl = h_ctrl;
    const int nb = h_nb;
    const bool helper = !PERSIST && (int) blockIdx.x == h_grid - 1 && (int) blockIdx.x >= nb;  // (persistent loop: persist_helper)
    // XCD-aware tile mapping.  Workgroups go round-robin to the 8 XCDs (workgroup b -> XCD b % 8) and every XCD has an L2
    // of its own, so with tile = workgroup the 256 particles next to a tile always belong to another XCD: after a resample
    // the ancestor's pose, genealogy and records -- written one launch ago by a neighbouring tile -- missed this XCD's L2
    // and came over the fabric (pose level +2.2 us against +0.6 us for the particle's own slot,
    // profiles/update_kernel_levels_r02_mid_N100000.txt).  Tile bt = the j-th tile of XCD x's contiguous range of the
    // particle set instead: stratified ancestors are near i, so they were written by this XCD.
    // Compact contexts: the packet rides in the kernel-argument segment (no staging copy on the stream).  Reading it there
    // element by element (U.small.idf[k] ...) means a scalar load per access into a scalar cache that is cold at every
    // launch: 18 % of the kernel's scalar requests waited on a miss (SQC_DCACHE_MISSES + _DUPLICATE,
    // profiles/rocprof_sq_counters_r02_c3_*.txt), ~14 per wave, each a trip to L2 / memory in the middle of the dependent
    // chain.  Instead: ONE coalesced vector load of the whole struct at kernel entry, in flight together with the block
    // totals, parked in LDS; the per-landmark reads below are LDS broadcasts.
    if constexpr (DIST) {
        // Folded collective: the barrier between the previous launch and this one, inside this one.  The launch itself is
        // the statement "my previous launch has completed and released its results" (stream order); the first wave of the
        // first block passes it on to every peer (system-scope release store of the sequence number into the peer's flag word for this shard)
        // and polls this shard's own flag words until every peer has said the same about ITS previous launch; then it
        // opens the go word, which every block of this launch polls before it requests anything: nothing a peer
        // still reads is overwritten, nothing a peer has not finished writing is read (the block totals pushed into this
        // shard's table included).  Flag words are fine-grained memory; every spin is bounded and a time-out is reported.
        if (U.fold_seq != 0) {
            uint32_t *fl = B.peers[B.shard].flags;
            if (blockIdx.x == 0 && threadIdx.x < kWave) {  // (the first block to be dispatched: its first wave is the envoy)
                const int t = threadIdx.x;
                if (t < B.n_shards && t != B.shard) {
                    // (relaxed: what this store announces -- the previous launch's results -- was released by that launch's
                    // end, and the launch boundary orders this store after it; a release here would only write back an L2
                    // that holds nothing dirty yet, for a microsecond)
                    __hip_atomic_store(B.peers[t].flags + B.shard, U.fold_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    uint32_t spins = 0;
                    while ((int32_t) (__hip_atomic_load(fl + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - U.fold_seq) < 0) {
                        if (__hip_atomic_load(fl + kMaxShards, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0 || ++spins > U.fold_spins) {
                            __hip_atomic_store(fl + kMaxShards, U.fold_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                // the wave has reconverged: every lane's peer has arrived or been given up on.  ONE acquire at system scope
                // by this wave (its CU's vector cache, and whatever the XCD's L2 holds of peer-written lines), so that what the
                // peers stored before their announcement is what this launch reads: one wave's buffer_inv, not hundreds of
                // blocks'; the go-word polls of the other blocks stay relaxed
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (threadIdx.x < kGoWords)
                    __hip_atomic_store(fl + kGoBase + kGoStride * threadIdx.x, U.fold_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            } else if (threadIdx.x == 0) {
                uint32_t spins = 0;
                const uint32_t *go = fl + kGoBase + kGoStride * (blockIdx.x % kGoWords);
                while ((int32_t) (__hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - U.fold_seq) < 0) {
                    if (++spins > 4u * U.fold_spins) {
                        __hip_atomic_store(fl + kMaxShards, U.fold_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                // (no cache invalidation here: caches were invalidated when this launch started and nothing has been read
                // since, so no wave can hold a stale line of what the peers wrote meanwhile; the go word itself is read past
                // the caches.  Hundreds of blocks invalidating the L2 at once cost more than the whole barrier.)
                asm volatile("" ::: "memory");
            }
            __syncthreads();
        }
    }
    // HEAD: everything whose address follows from the preloaded arguments is requested now, in one burst
    const bool front = PERSIST || (!BIG && MODE == 0 && (h_flags & 16) != 0);
    FrontLm f_lm{-1, 0};
    FrontHdr f_hd{0, -1, 0, 0};
    float f_x = 0.f, f_y = 0.f;
    // dword offsets in the kernel-argument segment (40: the head)
    constexpr size_t ka0 = (40 + sizeof(Buffers) + alignof(PredictArgs) - 1) / alignof(PredictArgs) * alignof(PredictArgs);
    constexpr size_t ka1 = (ka0 + sizeof(PredictArgs) + alignof(UpdateArgs) - 1) / alignof(UpdateArgs) * alignof(UpdateArgs);
    constexpr size_t ka_small = (ka1 + offsetof(UpdateArgs, small)) / 4;
    if constexpr (!BIG && MODE == 0 && !PERSIST) {  // (persistent loop: the helper workgroup makes the packets, an iteration ahead)
        if (front && threadIdx.x < kWave) {
            // front-end launches: this wave's oldest loads: the map (kernel arguments), then the state the previous launch left
            const auto *kf = (const __attribute__((address_space(4))) float *) __builtin_amdgcn_kernarg_segment_ptr();
            const int t = min((int) threadIdx.x, kSmallObs - 1);
            f_x = kf[ka_small + offsetof(SmallObs, zn) / 4 + t];
            f_y = kf[ka_small + offsetof(SmallObs, zn) / 4 + kSmallObs + t];
            f_hd = h_front->hdr;
            f_lm = h_front->lm[threadIdx.x];
        }
    }
    const bool logw = (h_flags & 4) != 0;
    const bool do_scan = h_plan && !h_scan_global;
    ScanLoads scl{0.0f, 0.0f, 0.0f, 0.0f, -INFINITY, -INFINITY};
    // distributed contexts, table wider than two totals per thread: by LDS-DMA into the memory of the ancestor windows and the
    // landmark staging (both used only after the scan), when the table fits there
    // (the launcher knows whether it fits: h_flags bit 5 -- a head argument, like everything the scan's requests depend on)
    float *scan_tab = nullptr;
    if constexpr (DIST) {
        if (do_scan && (h_flags & 32) != 0) scan_tab = wins;
    }
    // (persistent loop: the table is the one the PREVIOUS iteration wrote: by parity, not by the launch's leading argument)
    const float *__restrict__ tot = PERSIST ? (STEP_WPAR ? ws.blk_w[0] : ws.blk_w[1]) : h_tot;
    if (do_scan) {
        if (scan_tab) scan_issue_dma(tot, nbg, h_nb, scan_tab);
        else if (PERSIST && !logw) scl = scan_issue_small(tot, nbg);
        else scl = scan_issue<PERSIST>(tot, nbg, h_nb, logw);
    }
    // persistent loop, at most kWinBlocks tiles: every wave requests ALL in-block prefixes of the previous iteration now (1 KB per
    // tile, one 16-byte load per lane and tile) and parks them in its ancestor window behind the scan: a resampling iteration
    // finds its ancestors without another trip (the per-step launch tried the same and lost: its rows are the last thing the
    // PREVIOUS LAUNCH wrote, several microseconds away at the head of a launch; here they are L2 hits)
    const bool pre_win = PERSIST && do_scan && nbg <= kWinBlocks;
    float4 pw0 = make_float4(0.f, 0.f, 0.f, 0.f), pw1 = pw0, pw2 = pw0, pw3 = pw0;
    if constexpr (PERSIST) {
        if (pre_win) {
            const float *__restrict__ lc = STEP_WPAR ? ws.lcum[0] : ws.lcum[1];
            const int ln = threadIdx.x & (kWave - 1);
            pw0 = ldg<true>(reinterpret_cast<const float4 *>(lc) + ln);
            if (nbg > 1) pw1 = ldg<true>(reinterpret_cast<const float4 *>(lc + kBlock) + ln);
            if (nbg > 2) pw2 = ldg<true>(reinterpret_cast<const float4 *>(lc + 2 * kBlock) + ln);
            if (nbg > 3) pw3 = ldg<true>(reinterpret_cast<const float4 *>(lc + 3 * kBlock) + ln);
        }
    }
    // ... and, FastSLAM 1 in the fast build: the (V, G) normals of this particle's eight predicts, made an iteration ahead by a
    // drawer workgroup (persist_draw): four 16-byte loads in flight with the totals instead of ~1.9 us of Philox + Box-Muller
    float4 dq0 = make_float4(0.f, 0.f, 0.f, 0.f), dq1 = dq0, dq2 = dq0, dq3 = dq0, dq4 = dq0, dq5 = dq0;
    bool drawn = false;
    if constexpr (PERSIST) {
        drawn = carry.draw_src != nullptr && persist_batch_draws(METHOD, PA) && bid < nb;
        if (drawn) {
            const size_t Sd = (size_t) B.ncap, at = (size_t) bid * kBlock + threadIdx.x;
            dq0 = ldg<true>(carry.draw_src + at);
            dq1 = ldg<true>(carry.draw_src + Sd + at);
            dq2 = ldg<true>(carry.draw_src + 2 * Sd + at);
            dq3 = ldg<true>(carry.draw_src + 3 * Sd + at);
            dq4 = ldg<true>(carry.draw_src + 4 * Sd + at);
            dq5 = ldg<true>(carry.draw_src + 5 * Sd + at);
        }
    }
    __shared__ int32_t pk[kSmallWords];
    __shared__ uint32_t f_sets[4];
    __shared__ float f_aux[2 * kWave + 2];
    int32_t pkv = 0;
    // ... and the queued controls (PredictArgs::steps: 16 x 8 dwords) the same way, for the predict loop (compact contexts)
    constexpr int kStepWords = (int) (sizeof(PredictStep) / 4) * kMaxFusedPredict;
    static_assert(sizeof(PredictStep) == 32 && kStepWords <= kBlock, "one dword of PredictArgs::steps per thread");
    __shared__ float sh_ctl[BIG ? 1 : kStepWords];
    float ctlv = 0.0f;
    if constexpr (!BIG && !PERSIST) {
        constexpr size_t at = ka_small;  // dword offset of U.small in the kernel arguments
        const auto *ka = (const __attribute__((address_space(4))) int32_t *) __builtin_amdgcn_kernarg_segment_ptr();
        if (!front && threadIdx.x < kSmallWords) pkv = ka[at + threadIdx.x];  // (parked in LDS below, once the scan's loads are out too)
        const auto *kc = (const __attribute__((address_space(4))) float *) __builtin_amdgcn_kernarg_segment_ptr();
        if (threadIdx.x < kStepWords) ctlv = kc[(ka0 + offsetof(PredictArgs, steps)) / 4 + threadIdx.x];
    }
    if constexpr (PERSIST) {
        // the iteration's packet: made by the helper workgroup during the previous iteration, in the XCD's L2
        if (threadIdx.x < kSmallWords) pkv = ldg<true>(carry.pk_src + threadIdx.x);
    }
    // (persistent loop: the queued controls are already in LDS, in the iteration's queue entry)
    const CtlP ctl = PERSIST ? (CtlP) reinterpret_cast<const float *>(qe->PA.steps) : (CtlP) sh_ctl;
    (void) ctl;  // (strict build: the predict loop reads PredictArgs itself)
    const int cur = PERSIST ? carry.cur : h_ctrl->live[h_slot];  // ... and the Ctrl words
    const bool pend_word = PERSIST ? carry.pend_word : h_ctrl->pend[h_slot] != 0;
    if constexpr (!BIG && MODE == 0 && !PERSIST) {
        // front-end launches: the packet is worked out here, while the scan's loads and the Ctrl words are in flight
        if (front && bid < h_nb) {
            const int fw = threadIdx.x / kWave, ft = threadIdx.x & (kWave - 1);
            FrontGeom fg{};
            if (fw == 0) {
                fg = front_geometry(STEP_FRONT, f_x, f_y);
            } else if (fw == 1) {
                front_draw(STEP_FRONT, ft, f_aux);
            } else if (fw == 2) {
                const float cph = cosf(STEP_FRONT.phi), sph = sinf(STEP_FRONT.phi);
                if (ft == 0) {
                    f_aux[2 * kWave] = cph;
                    f_aux[2 * kWave + 1] = sph;
                }
            } else {
                // (every word a host-made packet would carry is defined: loops further down read clamped entries past m and n)
                for (int w = ft; w < kSmallWords; w += kWave) pk[w] = 0;
            }
            if (threadIdx.x < kStepWords) sh_ctl[threadIdx.x] = ctlv;
            __syncthreads();
            if (fw == 0) {
                const auto *kf = (const __attribute__((address_space(4))) float *) __builtin_amdgcn_kernarg_segment_ptr();
                const FrontObs ob = front_observe(STEP_FRONT, fg, f_aux, kf + ka_small + offsetof(SmallObs, zf) / 4);
                front_book(STEP_FRONT, ob, f_lm, f_hd, pk, f_sets, blockIdx.x == 0);
            }
        }
    }
    int bt = bid;
    if (!PERSIST && bt < nb) {
        const int x = bt & 7, j = bt >> 3, q = nb >> 3, r = nb & 7;
        bt = x < r ? x * (q + 1) + j : r * (q + 1) + (x - r) * q + j;
    }
    SLAM_STAMP(1);  // Ctrl words arrived
    // Where does particle i of the set this update works on come from?
    //   plan_inline: the resampling stage of the previous update has not run: every block redoes its scan of the block
    //                totals (=> sum w, Neff, decision, identical everywhere) and every thread finds its own ancestor
    //                (core.cpp:718-749, :800-806); no resample: slot i, weight w / sum(w) (core.cpp:726-729);
    //   otherwise  : resample_kernel ran: slot keep[i] of the live buffers if it left a gather pending, else slot i.
    // Either way a gathered particle is written to slot i of the OTHER pose / genealogy buffers.
    bool pend = (h_flags & 8) && pend_word;
    double W = 1.0, Mx = 0.0;
    // large contexts: the prefix comes from scan_kernel (same function, same association, run once) instead of being
    // redone by every block -- O(N^2 / 65 536) otherwise
    const double *offp = h_scan_global ? ws.scan[STEP_WPAR ^ 1] : off;
    if (h_plan && !helper) {
        double Q;
        if (h_scan_global) {
            W = offp[nb + 1];
            Q = offp[nb + 2];
            Mx = offp[nb + 3];
        } else if (PERSIST && !logw) {
            scan_small(scl, nb, off, W, Q);  // (every wave for itself: no barrier)
        } else {
            scan_finish(scl, tot, nbg, nb, logw, off, sh_a, sh_q, W, Q, Mx, scan_tab);
        }
        const float neff = neff_of(W, Q);  // Neff = 1 / sum((w/W)^2)  (core.cpp:784-788)
        pend = U.do_resample && (neff < (float) U.n_effective);
        if (bid == 0 && threadIdx.x == 0) {
            ctrl->wsum = W;
            ctrl->wsq = Q;
            ctrl->wmax = Mx;
            ctrl->neff = neff;
            ctrl->resampled = pend ? 1 : 0;
            ctrl->status = weight_status(W, Q);
            ws.est_part[STEP_WPAR ^ 1][4 * (size_t) nb] = (double) neff;  // travels with the partials into the history
            ws.est_part[STEP_WPAR ^ 1][4 * (size_t) nb + 1] = (double) ((pend ? 1 : 0) | (weight_status(W, Q) << 1));
        }
        if (!(PERSIST && !logw)) __syncthreads();
    }
    SLAM_STAMP(2);  // block totals scanned: W, Neff, decision known
    const int out = pend ? cur ^ 1 : cur;
    if constexpr (PERSIST) {
        carry.cur = out;
        if (pre_win && pend) {  // (uniform)
            const int ln = threadIdx.x & (kWave - 1);
            reinterpret_cast<float4 *>(win)[ln] = pw0;
            reinterpret_cast<float4 *>(win + kBlock)[ln] = pw1;
            reinterpret_cast<float4 *>(win + 2 * kBlock)[ln] = pw2;
            reinterpret_cast<float4 *>(win + 3 * kBlock)[ln] = pw3;
            __builtin_amdgcn_wave_barrier();
        }
    }
    // (DIST: the GLOBAL index of the ancestor of local particle k; global particle ids key the strata)
    auto ancestor = [&](int k, bool valid) -> int {
        if (!STEP_PLAN) return valid ? ws.keep[B.slot][k] : 0;
        const int64_t gk = (int64_t) (DIST ? B.first : 0) + k;
        // (persistent loop: this thread's stratum was drawn while the workgroups were meeting: persist_predraw)
        const double target = valid ? (double) (PERSIST ? carry.strat : stratum_prev(rng, gk)) * W : 0.0;
        const int64_t ng = DIST ? rng.n_global : (int64_t) B.n;
        if (PERSIST && pre_win)
            return (int) find_ancestor_win<true, true, true>(target, valid, (int) (gk >> 8), offp, nbg, win, ws.lcum[STEP_WPAR ^ 1], nb, ng,
                                                       logw ? ws.blk_w[STEP_WPAR ^ 1] + 2 * nb : nullptr, Mx, nullptr, STEP_WPAR ^ 1);
        // (the typed LDS pointer for the block prefix -- OFFL -- only in the persistent loop: 8.52 -> 8.29 us per iteration at config
        // 2; in the per-step kernels, which must keep the global-memory prefix of the largest contexts as well, the two copies of
        // the search cost more than the flat loads: config 3 14.02 -> 14.14 us per step, 10^6 particles 81.3 -> 85.3, same box)
        return (int) find_ancestor_win<false, PERSIST, PERSIST>(target, valid, (int) (gk >> 8), offp, nbg, win, ws.lcum[STEP_WPAR ^ 1], nb, ng,
                                                                (!DIST && logw) ? ws.blk_w[STEP_WPAR ^ 1] + 2 * nb : nullptr, Mx, DIST ? B.peers : nullptr, STEP_WPAR ^ 1);
    };
    if (bid >= nb) {
        // ---- helper blocks ---------------------------------------------------------------------------------
        if (helper) {
            // the set this launch leaves lives in `out` (published in the other Ctrl slot; the host flips after the
            // launch), the landmark rows' live flags for the next launch, and the pose estimate of an EARLIER update
            // whose partials are complete: this block runs beside the compute blocks instead of as launches of its own
            if (STEP_PLAN) {  // the decision is needed for `out`: recompute it from the two totals, cheaply
                double Q;
                if (U.scan_global) {
                    W = offp[nb + 1];
                    Q = offp[nb + 2];
                } else {
                    scan_finish(scl, tot, nbg, nb, logw, off, sh_a, sh_q, W, Q, Mx, scan_tab);
                }
                pend = U.do_resample && (neff_of(W, Q) < (float) U.n_effective);
            }
            if (threadIdx.x == 0) {
                ctrl->live[B.slot ^ 1] = pend ? cur ^ 1 : cur;
                ctrl->pend[B.slot ^ 1] = 0;
            }
            if (U.finalize) finish_estimate(B, ws, U.finalize_par, U.finalize_hist, sh_est);
            return;
        }
        if constexpr (BIG) {
            if (pend) {
                const PacketView V = packet_view(U);
                copy_genealogy(B, V.rows, V.n_rows, V.rows_per_role, ws, cur, U.copy_lo + bid - nb, ancestor);
            }
        }
        return;
    }
    if constexpr (!BIG) {
        // (parking these in front of the scan's last barrier, to save this one, was measured in round 4: config 2 +-0, config 3
        // 14.86 against 14.35 us per step -- the scan would then wait for the packet's loads as well)
        if ((PERSIST || !front) && threadIdx.x < kSmallWords) pk[threadIdx.x] = pkv;
        if (!front && threadIdx.x < kStepWords) sh_ctl[threadIdx.x] = ctlv;
        __syncthreads();
        if (!PERSIST && (uint32_t) pk[offsetof(SmallObs, magic) / 4] != kSmallMagic) {  // (layout guard: never seen)
            if (blockIdx.x == 0 && threadIdx.x == 0) ctrl->status = kStatusBadPacket;
            return;
        }
    }
    if constexpr (PP) {
        if (ppa.z_lds) {  // (uniform; every thread of a compute block gets here)
            for (int t = threadIdx.x; t < 2 * ppa.nz; t += kBlock) shZ[t] = ppa.z[t];
            __syncthreads();
        }
    }
    const int i = bt * kBlock + threadIdx.x;
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    // select (not index) the buffers: an indexed read of the pointer table in the kernel-argument segment
    // would be one more dependent scalar load at the head of every wave
    float4 *__restrict__ poseAo = out ? B.poseA[1] : B.poseA[0];
    float4 *__restrict__ poseBo = out ? B.poseB[1] : B.poseB[0];
    float2 *__restrict__ poseCo = out ? B.poseC[1] : B.poseC[0];
    const bool active = i < B.n;
    int m = U.m, n = U.n, nf = U.nf, e_new = U.e_new;
    int live_chunks = U.live_chunks, n_cons = U.n_cons;
    bool all_fresh = U.all_fresh != 0;
    if constexpr (!BIG && MODE == 0) {
        if (front) {  // the header front_make left in LDS (uniform: kept in scalar registers)
            const int32_t *h = pk + offsetof(SmallObs, head) / 4;
            m = __builtin_amdgcn_readfirstlane(h[kFrontHeadM]);
            n = __builtin_amdgcn_readfirstlane(h[kFrontHeadN]);
            nf = __builtin_amdgcn_readfirstlane(h[kFrontHeadNf]);
            e_new = __builtin_amdgcn_readfirstlane(h[kFrontHeadENew]);
            live_chunks = __builtin_amdgcn_readfirstlane(h[kFrontHeadChunks]);
            all_fresh = __builtin_amdgcn_readfirstlane(h[kFrontHeadFresh]) != 0;
            n_cons = __builtin_amdgcn_readfirstlane(h[kFrontHeadCons]);
        }
    }
    PacketView PV{};
    if constexpr (BIG) {
        PV = packet_view(U);
        m = PV.m;
        n = PV.n;
        nf = PV.nf;
        e_new = PV.e_new;
        n_cons = PV.n_cons;
    }
    float w = logw ? -INFINITY : 0.0f;  // lanes beyond the particle count carry no weight

    EstItem ei_prev{0.0, 0.0, -3.0e38f, 0.0f, 0x7fffffff};  // inline plan: this particle's term of the previous step's estimate
    const int si_plan = pend ? ancestor(i, active) : i;  // (every lane of the block: the windowed search is a wave's joint effort)
    if (active) {
        // where this particle's pose and genealogy are read from: slot i of the live buffers, or its ancestor's slot; or
        // (sharded runs, keep[i] < 0) slot i of the OUTPUT buffers: the particle arrived from another shard and
        // shard_unpack_kernel has already put its pose and genealogy in place
        int si = si_plan;
        SLAM_STAMP(3);  // ancestor found (two dependent rounds of the in-block search)
        int sb = cur;
        if (ARR && si < 0) {
            si = i;
            sb = out;
        }
        // DIST: `si` is a global index when a resample is applied: which shard owns it, and where
        int gsrc = B.first + i;          // global id of the source slot
        bool src_local = true;
        const PeerPtrs *rp = B.peers;    // the owning shard's arrays (only read when the source is remote; read from the
                                         // table in place: a private copy indexed by `sb` would live in scratch)
        if (DIST && pend) {
            gsrc = si;
            const int h = (int) __umul64hi((unsigned long long) (unsigned) si, B.div_n);
            si -= h * B.ncap;
            src_local = h == B.shard;
            rp = B.peers + h;
            // (bookkeeping for the bench line: how much of a resample crosses xGMI)
            const unsigned long long rm = __ballot(!src_local);
            if ((h_flags & 64) && rm && lane == (int) __ffsll((long long) __ballot(true)) - 1) atomicAdd(&ctrl->remote_reads, (unsigned long long) __popcll(rm));
        }
        const float4 *__restrict__ poseA = (DIST && !src_local) ? rp->poseA[sb] : (sb ? B.poseA[1] : B.poseA[0]);
        const float4 *__restrict__ poseB = (DIST && !src_local) ? rp->poseB[sb] : (sb ? B.poseB[1] : B.poseB[0]);
        const float2 *__restrict__ poseC = (DIST && !src_local) ? rp->poseC[sb] : (sb ? B.poseC[1] : B.poseC[0]);
        const int32_t *__restrict__ genS = (DIST && !src_local) ? rp->gen[sb] : (sb ? B.gen[1] : B.gen[0]);
        int32_t *__restrict__ genO = out ? B.gen[1] : B.gen[0];
        struct Rec {
            float4 a;
            float b;
        };
        // (the four record-buffer pointers as plain values: selecting between B.lmkA[0] and B.lmkA[1] in place made the
        // distributed variants index a private copy of the argument struct: 40 bytes of scratch per lane and its set-up)
        float4 *const lmkA0 = B.lmkA[0], *const lmkA1 = B.lmkA[1];
        float *const lmkB0 = B.lmkB[0], *const lmkB1 = B.lmkB[1];
        // record of landmark j in slot s of record buffer b (b: bit 30 of the packet's row word, see slot_of / buf_of)
        auto load_rec = [&, lmkA0, lmkA1, lmkB0, lmkB1](int j, int s, int b) -> Rec {
            // kPoolBit set: a record that arrived from another shard lives in the arrival pool (kernels.h: Buffers::poolA).
            // Address select, not a branch: the staging arrays these references point into must stay in registers.
            const bool pool = ARR && s < 0;
            size_t at = pool ? (size_t) j * B.pool_cap + (size_t) (s & ~kPoolBit) : (size_t) j * S + (size_t) s;
            const float4 *pA = pool ? B.poolA : (b ? lmkA1 : lmkA0);
            const float *pB = pool ? B.poolB : (b ? lmkB1 : lmkB0);
            if (DIST) {  // s is a global slot id: almost always one of this shard's
                const int ls = s - B.first;
                if (ls >= 0 && ls < B.ncap) {
                    at = (size_t) j * S + (size_t) ls;
                } else {
                    const int h = (int) __umul64hi((unsigned long long) (unsigned) s, B.div_n);
                    at = (size_t) j * S + (size_t) (s - h * B.ncap);
                    pA = B.peers[h].lmkA[b];
                    pB = B.peers[h].lmkB[b];
                    // (a value, not a load the optimiser may sink below the join: it did, by parking the LOCAL pointers in a
                    // private array so that both arms became "load a pointer from memory": 40 bytes of scratch per lane)
                    asm volatile("" : "+v"(pA), "+v"(pB));
                }
            }
            return Rec{ldg<PERSIST>(pA + at), ldg<PERSIST>(pB + at)};  // by value: a reference into the staging arrays would pin them to scratch
        };
        auto load_lmk = [&](int j, int s, int b, float4 &la, float &lb) {
            const Rec r = load_rec(j, s, b);
            la = r.a;
            lb = r.b;
        };
        // a re-observed landmark's fresh record goes to the particle's own slot of the row's OTHER buffer
        auto store_lmk = [&, lmkA0, lmkA1, lmkB0, lmkB1](int j, int b, const float4 &la, float lb) {
            nt_store(&(b ? lmkA0 : lmkA1)[(size_t) j * S + i], la);
            nt_store(&(b ? lmkB0 : lmkB1)[(size_t) j * S + i], lb);
        };
        auto store_new = [&, lmkA0, lmkB0](int j, const float4 &la, float lb) {  // a new row starts in record buffer 0 (host: live flag 0)
            nt_store(&lmkA0[(size_t) j * S + i], la);
            nt_store(&lmkB0[(size_t) j * S + i], lb);
        };

        // BIG: the packet sits in device memory, written before the launch and never during it: read it through the
        // constant address space, so that the (uniform) per-landmark reads become scalar loads.  As plain global pointers
        // they were VECTOR loads, and the s_waitcnt vmcnt(0) in front of their first use also waited for every record
        // prefetch and record store in flight: the software pipeline below never overlapped anything
        // (profiles/update_kernel_levels_r02_c5_*.txt: 0.78 + 1.18 us per landmark before, both passes).
        using IdxP = std::conditional_t<BIG, const __attribute__((address_space(4))) int32_t *, const int32_t *>;
        using FltP = std::conditional_t<BIG, const __attribute__((address_space(4))) float *, const float *>;
        IdxP idf, lrow;
        FltP zf, zn;
        if constexpr (BIG) {
            idf = (IdxP) reinterpret_cast<uintptr_t>(PV.idf);
            zf = (FltP) reinterpret_cast<uintptr_t>(PV.zf);
            zn = (FltP) reinterpret_cast<uintptr_t>(PV.zn);
            lrow = (IdxP) reinterpret_cast<uintptr_t>(PV.row);
        } else {
            idf = pk + offsetof(SmallObs, idf) / 4;
            lrow = pk + offsetof(SmallObs, row) / 4;
            zf = reinterpret_cast<const float *>(pk + offsetof(SmallObs, zf) / 4);
            zn = reinterpret_cast<const float *>(pk + offsetof(SmallObs, zn) / 4);
        }
        // re-observed landmark k of this particle: the slot comes from the genealogy row the landmark uses (kernels.h:
        // gen), the buffer from the row's live flag; a landmark this update writes goes to the particle's OWN slot of the
        // row's other buffer and into the genealogy row this update opens (U.e_new: identity)
        // (compact contexts = small packets: rows interleaved four to a chunk, kernels.h: Buffers::gen)
        // (DIST: genealogy entries and the values returned here are GLOBAL slot ids)
        auto slot_of = [&](int k) -> int {
            return (lrow[k] & kRowFreshBit) ? (DIST ? gsrc : si) : ldg<PERSIST>(genS + gen_index(!BIG, S, lrow[k] & kRowMask, (size_t) si));
        };
        auto buf_of = [&](int k) -> int { return (lrow[k] >> 30) & 1; };
        const float r00 = U.R[0], r01 = U.R[1], r10 = U.R[2], r11 = U.R[3];
        // Per-particle association (PP; kernels.h: PerParticle): packet entry k is a landmark SOME particle matched; this particle's
        // observation of it is pp_j(k) (-1: it leaves the landmark alone and copies its record forward).  pp_any: bit 0 it matched a
        // landmark, bit 1 it opens one; neither: the step leaves its pose and Pv alone (a particle without an observation is not
        // sampled: fastslam2.cpp:21-48 runs only for a non-empty z).
        [[maybe_unused]] int pp_any = 3;
        if constexpr (PP) pp_any = (int) ppa.any[i];
        // (re-observed entries, k < m: staged by the pipeline with the records, one chunk ahead; the new slots behind them: read in place)
        [[maybe_unused]] auto pp_j = [&](int k) -> int {
            return k < m ? shJ[(k & (kBigChunk - 1)) * kBlock + threadIdx.x] : (int) ppa.obs[(size_t) k * S + i];
        };
        [[maybe_unused]] auto pp_z = [&](int t) -> float { return ppa.z_lds ? shZ[t] : ppa.z[t]; };
        // Stage the first KS re-observed landmarks in LDS with all their loads in flight together (one HBM latency instead
        // of one per landmark); both passes then read LDS.  Measured before this: 38 % of the wave's cycles were s_waitcnt
        // stalls (profiles/rocprof_sq_counters_r01.txt).  Unconditional loads (index clamped to the last landmark): no
        // branch between them, so the compiler issues all of them before the first s_waitcnt; duplicates are L1 hits.
        // Two sizes: most steps re-observe at most kStage/2 landmarks and need not pay for eight address computations.
        // Split in two so that the loads can be requested as early as their addresses are known (with the pose when every
        // staged landmark is fresh: kRowFreshBit) and the wave only waits for them when it needs them.
        float4 sta[kStage];
        float stb[kStage];
        auto issue_records = [&](auto KS, const int *ts) {
            constexpr int ks = decltype(KS)::value;
#pragma unroll
            for (int k = 0; k < ks; k++) {
                const Rec r = load_rec(idf[min(k, m - 1)], ts[k], buf_of(min(k, m - 1)));
                sta[k] = r.a;
                stb[k] = r.b;
            }
        };
        auto commit_records = [&](auto KS) {
            constexpr int ks = decltype(KS)::value;
#pragma unroll
            for (int k = 0; k < ks; k++) {
                shA[(k) * kBlock + threadIdx.x] = sta[k];
                shB[(k) * kBlock + threadIdx.x] = stb[k];
            }
        };
        auto stage_landmarks = [&](const int *ts, bool issued) {
            if (m <= kStage / 2) {
                if (!issued) issue_records(std::integral_constant<int, kStage / 2>{}, ts);
                commit_records(std::integral_constant<int, kStage / 2>{});
            } else {
                if (!issued) issue_records(std::integral_constant<int, kStage>{}, ts);
                commit_records(std::integral_constant<int, kStage>{});
            }
        };

        // BIG: body(k, la, lb) for k = 0..m-1, in order, over the re-observed landmarks, as a three-stage software pipeline
        // (see kBigChunk).  Out-of-range tail entries re-read the last landmark (never consumed).
        auto pipeline = [&](auto body) {
            constexpr int CH = kBigChunk;
            int sl[CH];
            float4 ta[CH];
            float tb[CH];
            [[maybe_unused]] int tj[CH];
            static_assert(CH == kBigChunk && (kBigChunk & (kBigChunk - 1)) == 0, "pp_j indexes the staged chunk by k mod kBigChunk");
            auto load_slots = [&](int k0) {
#pragma unroll
                for (int k = 0; k < CH; k++) sl[k] = slot_of(min(k0 + k, m - 1));
            };
            auto load_recs = [&](int k0) {
#pragma unroll
                for (int k = 0; k < CH; k++) {
                    const Rec r = load_rec(idf[min(k0 + k, m - 1)], sl[k], buf_of(min(k0 + k, m - 1)));
                    ta[k] = r.a;
                    tb[k] = r.b;
                    if constexpr (PP) tj[k] = (int) ppa.obs[(size_t) min(k0 + k, m - 1) * S + i];
                }
            };
            load_slots(0);
            load_recs(0);
            load_slots(CH);
            for (int k0 = 0; k0 < m; k0 += CH) {
#pragma unroll
                for (int k = 0; k < CH; k++) {
                    shA[(k) * kBlock + threadIdx.x] = ta[k];
                    shB[(k) * kBlock + threadIdx.x] = tb[k];
                    if constexpr (PP) shJ[(k) * kBlock + threadIdx.x] = tj[k];
                }
                if (k0 + CH < m) {
                    load_recs(k0 + CH);       // slots of this chunk were requested one chunk of compute ago
                    load_slots(k0 + 2 * CH);  // (clamped: harmless re-reads past the end)
                }
                const int kn = min(CH, m - k0);
                for (int k = 0; k < kn; k++) body(k0 + k, shA[(k) * kBlock + threadIdx.x], shB[(k) * kBlock + threadIdx.x]);
            }
        };

        // Compact contexts: a pending gather's genealogy composition (gen_out[e][i] = gen[e][ancestor]) is done right here
        // by the particle's own thread -- it knows its ancestor already -- instead of by helper blocks that would each redo
        // the scan and the search: all of the particle's chunks (four rows each, at most kSmallRows / 4) requested with the
        // pose, stored when it arrives, with the row this update opens already set to "own slot".
        constexpr int kChunks = (kSmallRows + 3) / 4;
        const bool copy_inline = !BIG && pend && sb == cur;  // (an arrival's genealogy is already in place)
        const int nchunks = live_chunks;  // chunks holding a row in use after this update (incl. the one it opens)
        int4 gq[kChunks];
        float4 pa = ldg<PERSIST>(poseA + si);
        // FastSLAM 2: the pose covariance with the pose, and consumed (below) BEFORE the genealogy chunks of a pending gather are
        // requested: requested behind them, as the source order had it until round 5, it came back behind them -- loads return in
        // order -- and the predicts, the first to need it, waited a second trip on every launch and for all the chunks on a
        // resampling one (undrained level stamps, profiles/update_kernel_flow_r05_*.txt: pose -> predicts +1.47 us on a resampling
        // launch of config 3 against +0.68 otherwise, +5.8 against +3.0 at config 6, whose 117 landmarks keep five chunks alive).
        float4 pb = make_float4(0.f, 0.f, 0.f, 0.f);
        float2 pc = make_float2(0.f, 0.f);
        if (METHOD == 2) {
            pb = ldg<PERSIST>(poseB + si);
            pc = ldg<PERSIST>(poseC + si);
        }
        // the slots of the (first kStage) re-observed landmarks are fetched now, with the pose: they depend on nothing but
        // the source slot, so the records are one round trip behind the pose, not two
        int ts[kStage];
        const bool early_records = !BIG && m > 0 && all_fresh;
        if (!BIG) {
#pragma unroll
            for (int k = 0; k < kStage; k++) ts[k] = all_fresh ? (DIST ? gsrc : si) : slot_of(min(k, max(m - 1, 0)));
            if (early_records) {  // fresh landmarks: the record sits in the source slot: requested with the pose
                if (m <= kStage / 2) issue_records(std::integral_constant<int, kStage / 2>{}, ts);
                else issue_records(std::integral_constant<int, kStage>{}, ts);
            }
        }
        // the particle's normals (device draws) are worked out HERE, while the pose is in flight: ~150 instructions that need
        // nothing from memory and used to run after the pose had arrived (15.17 -> 14.62 us per step, measured)
        float hg0 = 0.f, hg1 = 0.f, hg2 = 0.f;
        if constexpr (PERSIST) {  // (drawn while the workgroups were meeting: persist_predraw)
            hg0 = carry.hg0;
            hg1 = carry.hg1;
            hg2 = carry.hg2;
        } else if (METHOD == 2 && rng.mode != 0 && (m > 0 || n > 0)) {
            U4 r = philox4x32((uint32_t) (rng.first_particle + i), rng.step, 0u, 0u, rng.k0, rng.k1);
#ifdef SLAM_FAST_MATH
            box_muller3_fast(r, hg0, hg1, hg2);
#else
            box_muller3(r, hg0, hg1, hg2);
#endif
            asm volatile("" : "+v"(hg0), "+v"(hg1), "+v"(hg2));  // (pinned above the wait for the pose)
        }
#ifdef SLAM_FAST_MATH
        // FastSLAM 1, the usual five to eight queued predicts: their (V, G) normals depend on counters only and are drawn HERE,
        // in one batch, while the pose is in flight (~1 us of arithmetic behind a ~0.9 us trip; predict_steps_fs1_fast)
        constexpr int kEarly = 8;
        float pg0[kEarly], pg1[kEarly];
        const bool early_draws = METHOD == 1 && !BIG && PA.nsteps > kEarly / 2 && PA.nsteps <= kEarly && PA.add_noise && !PA.use_heading && !PA.comp.valid;
        if (PERSIST && drawn) {
            // (the pose-free half of the predicts came ready-made: applied below)
        } else if (early_draws) {
            draw_batch_fs1_fast<kEarly>(pg0, pg1, PA, rng, i, S, ctl, 0, PA.nsteps);
#pragma unroll
            for (int q = 0; q < kEarly; q++) asm volatile("" : "+v"(pg0[q]), "+v"(pg1[q]));  // (pinned above the wait for the pose)
        }
#endif
        SLAM_STAMP(4);  // pose + genealogy of the ancestor arrived
        float x = pa.x, y = pa.y, th = pa.z;
        if (METHOD == 2) asm volatile("" : "+v"(pb.x), "+v"(pb.y), "+v"(pb.z), "+v"(pb.w), "+v"(pc.x), "+v"(pc.y));  // (arrived: see above)
        if (!BIG && copy_inline) {
            // requested only now, behind the pose and the records on the in-order return path (the step's critical chain);
            // they arrive during the compute below and are stored with the pose at the end
            // UNGUARDED loads, in a few sizes behind a uniform branch (a chunk past the last live one re-reads that one: the same 16
            // bytes again).  Guarded one by one (`if (c < nchunks)`, until round 5) every load became a branch, a load and an
            // s_waitcnt vmcnt(0) -- dependent round trips, two or three on a resampling launch of example_webmap, five on one of
            // example_loop902 (undrained level stamps: pose -> predicts +6.1 us there against +3.0 on a launch that does not
            // resample).  Which sizes: where the sizes' registers meet again the compiler copies some of them, and a copy waits for
            // its load; measured, one box, us per step: sizes 2 / 4 / 7 / 10 everywhere: config 3 13.9-14.0, config 6 16.7, config 2's
            // loop 8.51, 10^6 particles 82.8-83.2; all ten always: 14.0, 16.2, 8.81 (its loads pass the vector cache), 83.5-84.2.
            const int4 *__restrict__ g4 = reinterpret_cast<const int4 *>(genS);
            const int last = max(nchunks - 1, 0);
            auto load_chunks = [&](auto NB) {
#pragma unroll
                for (int c = 0; c < decltype(NB)::value; c++) gq[c] = ldg<PERSIST>(g4 + ((size_t) min(c, last) * S + si));
            };
            if (PERSIST) {
                if (nchunks <= 2) load_chunks(std::integral_constant<int, 2>{});
                else if (nchunks <= 4) load_chunks(std::integral_constant<int, 4>{});
                else if (nchunks <= 7) load_chunks(std::integral_constant<int, (kChunks < 7 ? kChunks : 7)>{});
                else load_chunks(std::integral_constant<int, kChunks>{});
            } else {
                if (nchunks <= 4) load_chunks(std::integral_constant<int, 4>{});
                else load_chunks(std::integral_constant<int, kChunks>{});
            }
        }
        // resampled particles restart at 1/N (core.cpp:744-747); otherwise the weights are normalised (core.cpp:726-729;
        // resample_kernel has already done it unless this launch plans inline)
        // (log-weight contexts: l - (M + log sum exp(l - M)); Ctrl.inv_n holds log(1/N))
        w = pend ? ctrl->inv_n : (STEP_PLAN ? (logw ? pa.w - (float) (Mx + log(W)) : pa.w / (float) W) : pa.w);
        // computeEstimatedPosition of the previous update (ParticleSLAMWrapper.cpp:56-77) sees exactly this set
        ei_prev = EstItem{(double) pa.x, (double) pa.y, w, pa.z, i};
        float q00 = 0.f, q10 = 0.f, q11 = 0.f, q20 = 0.f, q21 = 0.f, q22 = 0.f;
        bool pose_dirty = pend;
        if (METHOD == 2) {
            q00 = pb.x; q10 = pb.y; q11 = pb.z; q20 = pb.w; q21 = pc.x; q22 = pc.y;
        }
        if (PA.nsteps > 0) {
#ifdef SLAM_FAST_MATH
            if (PA.comp.valid) {
                Sym3 P = {q00, q10, q11, q20, q21, q22};
                predict_composite(x, y, th, P, PA.comp);
                q00 = P.p00; q10 = P.p10; q11 = P.p11; q20 = P.p20; q21 = P.p21; q22 = P.p22;
            } else if (METHOD == 2 && PA.use_heading && !PA.add_noise) {
                Sym3 P = {q00, q10, q11, q20, q21, q22};
                predict_steps_heading_fast(x, y, th, P, PA, BIG ? nullptr : ctl);
                q00 = P.p00; q10 = P.p10; q11 = P.p11; q20 = P.p20; q21 = P.p21; q22 = P.p22;
            } else if (METHOD == 1 && PA.add_noise && !PA.use_heading) {
                if (PERSIST && drawn) {
                    const float vd[kEarly] = {dq0.x, dq0.y, dq0.z, dq0.w, dq1.x, dq1.y, dq1.z, dq1.w};
                    const float gs[kEarly] = {dq2.x, dq2.y, dq2.z, dq2.w, dq3.x, dq3.y, dq3.z, dq3.w};
                    const float sgw[kEarly] = {dq4.x, dq4.y, dq4.z, dq4.w, dq5.x, dq5.y, dq5.z, dq5.w};
                    apply_controls_fs1_fast<kEarly>(x, y, th, vd, gs, sgw, PA.nsteps);
                } else if (early_draws) {
                    const L2 Lq = llt2(PA.Q[0], PA.Q[2], PA.Q[3]);
                    apply_batch_fs1_fast<kEarly>(x, y, th, pg0, pg1, PA, ctl, PA.dt, 1.0f / PA.wheel_base, Lq, 0, PA.nsteps);
                } else {
                    predict_steps_fs1_fast(x, y, th, PA, rng, i, S, BIG ? nullptr : ctl);
                }
            } else
#endif
            {
                float P[9] = {q00, q10, q20, q10, q11, q21, q20, q21, q22};
                predict_steps(x, y, th, P, PA, rng, i, S);
                q00 = P[0]; q10 = P[3]; q11 = P[4]; q20 = P[6]; q21 = P[7]; q22 = P[8];
            }
            pose_dirty = true;
        }
        SLAM_STAMP(10);  // queued predicts applied

#ifdef SLAM_FAST_MATH
        if (METHOD == 2) {
            // restructured arithmetic (device_math.h, fast section); same data flow as the strict branch below
            float g0 = 0.f, g1 = 0.f, g2 = 0.f;
            if (m > 0 || n > 0) {
                if (rng.mode == 0) {
                    g0 = rng.normals[0 * S + i];
                    g1 = rng.normals[1 * S + i];
                    g2 = rng.normals[2 * S + i];
                } else {
                    g0 = hg0;
                    g1 = hg1;
                    g2 = hg2;
                }
            }
            const float rl = 0.5f * (r01 + r10);
            if (m > 0) {
                const float x0 = x, y0 = y, th0 = th;
                Sym3 P = {q00, q10, q11, q20, q21, q22};
                const L3r L0 = llt3r(P);  // factor of Pv0 for the prior term (:366)
                auto first_pass = [&](int k, float4 la, float lb) {
                    if constexpr (PP) {
                        const int j = pp_j(k);
                        if (j < 0 || la.x != la.x) return;  // (not matched; or -- caller-made labels only -- a landmark this particle does not hold)
                        const Obs2 o = observe2(x, y, th, la.x, la.y, la.z, la.w, lb, r00, rl, r11);
                        proposal_update(x, y, th, P, o, pp_z(2 * j) - o.zp0, wrap_pi(pp_z(2 * j + 1) - o.zp1));
                    } else {
                        const Obs2 o = observe2(x, y, th, la.x, la.y, la.z, la.w, lb, r00, rl, r11);
                        proposal_update(x, y, th, P, o, zf[2 * k] - o.zp0, wrap_pi(zf[2 * k + 1] - o.zp1));
                    }
                };
                if constexpr (BIG) {
                    pipeline(first_pass);
                } else {
                    stage_landmarks(ts, early_records);
                    for (int k = 0; k < m; k++) {
                        float4 la;
                        float lb;
                        if (k < kStage) {
                            la = shA[(k) * kBlock + threadIdx.x];
                            lb = shB[(k) * kBlock + threadIdx.x];
                        } else {
                            load_lmk(idf[k], slot_of(k), buf_of(k), la, lb);
                        }
                        if (k == 0) SLAM_STAMP(5);  // records staged (slot -> record round trips done)
                        first_pass(k, la, lb);
                    }
                }
                SLAM_STAMP(6);  // proposal pass done
                const L3r Lp = llt3r(P);
                // (PP: a particle this step does not concern is not sampled)
                const float xs = (PP && pp_any == 0) ? x : ffma(Lp.l00, g0, x);
                const float ys = (PP && pp_any == 0) ? y : ffma(Lp.l11, g1, ffma(Lp.l10, g0, y));
                const float ths = (PP && pp_any == 0) ? th : ffma(Lp.l22, g2, ffma(Lp.l21, g1, ffma(Lp.l20, g0, th)));
                float lik = 1.0f;
                double dl = 0.0;  // log-weight contexts: sum of the log-likelihoods (double: ~1.3 k terms of ~4.5 at config 5)
                auto second_pass = [&](int k, float4 la, float lb) {
                    if constexpr (PP) {
                        const int j = pp_j(k);
                        if (j >= 0 && la.x == la.x) {
                            const Obs2 o = observe2(xs, ys, ths, la.x, la.y, la.z, la.w, lb, r00, rl, r11);
                            const Gauss2 g = feature_update2(la.x, la.y, la.z, la.w, lb, o, pp_z(2 * j) - o.zp0, wrap_pi(pp_z(2 * j + 1) - o.zp1));
                            if (logw) dl += (double) (g.E + __logf(g.norm));
                            else lik *= __expf(g.E) * g.norm;
                        }  // (else: not this particle's landmark in this step: the record moves on unchanged)
                        store_lmk(idf[k], buf_of(k), la, lb);
                    } else {
                        const Obs2 o = observe2(xs, ys, ths, la.x, la.y, la.z, la.w, lb, r00, rl, r11);
                        const Gauss2 g = feature_update2(la.x, la.y, la.z, la.w, lb, o, zf[2 * k] - o.zp0, wrap_pi(zf[2 * k + 1] - o.zp1));
                        if (logw) dl += (double) (g.E + __logf(g.norm));
                        else lik *= __expf(g.E) * g.norm;
                        store_lmk(idf[k], buf_of(k), la, lb);
                    }
                };
                if constexpr (BIG) {
                    pipeline(second_pass);
                } else {
                    const int ms = min(m, kStage);
                    // (two landmarks per basic block, as FastSLAM 1's pair_pass below, was measured here in round 4: 14.180
                    // against 14.184 us per step over 2 000 steps at 10^5 particles -- with ~1.5 waves per SIMD the other wave
                    // fills the dependent-issue gaps already)
                    for (int k = 0; k < ms; k++) second_pass(k, shA[(k) * kBlock + threadIdx.x], shB[(k) * kBlock + threadIdx.x]);
                    for (int k = ms; k < m; k++) {
                        float4 la;
                        float lb;
                        load_lmk(idf[k], slot_of(k), buf_of(k), la, lb);
                        second_pass(k, la, lb);
                    }
                }
                SLAM_STAMP(7);  // likelihood / feature-update pass done, record stores landed
                // w *= likelihood * prior / proposal (:360-367): one exponential for the ratio of the two Gaussians
                const float E = gauss3_exponent(L0, x0 - xs, y0 - ys, wrap_pi(th0 - ths)) -
                                gauss3_exponent(Lp, x - xs, y - ys, wrap_pi(th - ths));
                const float ratio = ((Lp.l00 * Lp.l11) * Lp.l22) * ((L0.r0 * L0.r1) * L0.r2);
                if constexpr (PP) {
                    if (pp_any & 1) {  // (no landmark matched: the proposal IS the prior, the two Gaussians cancel)
                        if (logw) w = (float) ((double) w + dl + (double) (E + __logf(ratio)));
                        else w = w * lik * (__expf(E) * ratio);
                    }
                    if (pp_any != 0) {
                        x = xs;
                        y = ys;
                        th = ths;
                        q00 = q10 = q11 = q20 = q21 = q22 = 0.0f;
                        pose_dirty = true;
                    }
                } else {
                    if (logw) w = (float) ((double) w + dl + (double) (E + __logf(ratio)));
                    else w = w * lik * (__expf(E) * ratio);
                    x = xs;
                    y = ys;
                    th = ths;
                    q00 = q10 = q11 = q20 = q21 = q22 = 0.0f;
                    pose_dirty = true;
                }
            } else if (n > 0 && (!PP || pp_any != 0)) {
                const L3r L = llt3r(Sym3{q00, q10, q11, q20, q21, q22});
                x = ffma(L.l00, g0, x);
                y = ffma(L.l11, g1, ffma(L.l10, g0, y));
                th = ffma(L.l22, g2, ffma(L.l21, g1, ffma(L.l20, g0, th)));
                q00 = q10 = q11 = q20 = q21 = q22 = 0.0f;
                pose_dirty = true;
            }
            for (int k = 0; k < n; k++) {
                float4 la;
                float lb;
                if constexpr (PP) {
                    const int j = pp_j(m + k);
                    if (j >= 0) {
                        add_feature_fast(x, y, th, pp_z(2 * j), pp_z(2 * j + 1), r00, r01, r10, r11, la.x, la.y, la.z, la.w, lb);
                    } else {  // the particle does not open this landmark: an absent record (kernels.h: kAbsent)
                        la = make_float4(kAbsent, kAbsent, 0.0f, 0.0f);
                        lb = 0.0f;
                    }
                    store_new(ppa.idn[k], la, lb);
                } else {
                    add_feature_fast(x, y, th, zn[2 * k], zn[2 * k + 1], r00, r01, r10, r11, la.x, la.y, la.z, la.w, lb);
                    store_new(nf + k, la, lb);
                }
            }
        } else
#endif
        if (METHOD == 2) {
            float g0 = 0.f, g1 = 0.f, g2 = 0.f;
            if (m > 0 || n > 0) {
                if (rng.mode == 0) {
                    g0 = rng.normals[0 * S + i];
                    g1 = rng.normals[1 * S + i];
                    g2 = rng.normals[2 * S + i];
                } else {
                    g0 = hg0;
                    g1 = hg1;
                    g2 = hg2;
                }
            }
            if (m > 0) {
                const float x0 = x, y0 = y, th0 = th;
                // running proposal covariance, full 3x3 (the reference's Pv stays a full matrix inside the loop)
                float P[9] = {q00, q10, q20, q10, q11, q21, q20, q21, q22};
                auto first_pass = [&](int k, float4 la, float lb) {
                    [[maybe_unused]] int jo = 0;  // (PP: this particle's observation of the landmark; the z reads below stay where they were)
                    if constexpr (PP) {
                        jo = pp_j(k);
                        if (jo < 0 || la.x != la.x) return;  // (not matched; or -- caller-made labels only -- a landmark this particle does not hold)
                    }
                    // Jacobians at the running mean (fastslam2.cpp:320,:348)
                    Jac j = jacobian(x, y, th, la.x, la.y, la.z, la.w, lb, r00, r01, r10, r11);
                    float s00, s01, s10, s11;
                    inverse2(j.s00, j.s01, j.s10, j.s11, s00, s01, s10, s11);  // Sfi (:324)
                    const float v0 = (PP ? pp_z(2 * jo) : zf[2 * k]) - j.zp0;
                    const float v1 = trig_offset((PP ? pp_z(2 * jo + 1) : zf[2 * k + 1]) - j.zp1);
                    float Pinv[9];
                    llt_solve_identity3(llt3(P[0], P[3], P[4], P[6], P[7], P[8]), Pinv);  // (:335)
                    // T1 = Hv^T * Sfi (3x2), T2 = T1 * Hv (3x3); Hv = [[hv00 hv01 0],[hv10 hv11 -1]]
                    const float t00 = j.hv00 * s00 + j.hv10 * s10, t01 = j.hv00 * s01 + j.hv10 * s11;
                    const float t10 = j.hv01 * s00 + j.hv11 * s10, t11 = j.hv01 * s01 + j.hv11 * s11;
                    const float t20 = -s10, t21 = -s11;
                    P[0] = (t00 * j.hv00 + t01 * j.hv10) + Pinv[0];
                    P[1] = (t00 * j.hv01 + t01 * j.hv11) + Pinv[1];
                    P[2] = (-t01) + Pinv[2];
                    P[3] = (t10 * j.hv00 + t11 * j.hv10) + Pinv[3];
                    P[4] = (t10 * j.hv01 + t11 * j.hv11) + Pinv[4];
                    P[5] = (-t11) + Pinv[5];
                    P[6] = (t20 * j.hv00 + t21 * j.hv10) + Pinv[6];
                    P[7] = (t20 * j.hv01 + t21 * j.hv11) + Pinv[7];
                    P[8] = (-t21) + Pinv[8];
                    llt_solve_identity3(llt3(P[0], P[3], P[4], P[6], P[7], P[8]), P);  // (:341)
                    // xv += ((Pv * Hv^T) * Sfi) * v   (:345)
                    float c[3];
#pragma unroll
                    for (int r = 0; r < 3; r++) {
                        const float a0 = P[3 * r] * j.hv00 + P[3 * r + 1] * j.hv01;
                        const float a1 = (P[3 * r] * j.hv10 + P[3 * r + 1] * j.hv11) + P[3 * r + 2] * -1.0f;
                        const float b0 = a0 * s00 + a1 * s10;
                        const float b1 = a0 * s01 + a1 * s11;
                        c[r] = b0 * v0 + b1 * v1;
                    }
                    x = x + c[0];
                    y = y + c[1];
                    th = th + c[2];
                };
                if constexpr (BIG) {
                    pipeline(first_pass);
                } else {
                    stage_landmarks(ts, early_records);
                    for (int k = 0; k < m; k++) {
                        float4 la;
                        float lb;
                        if (k < kStage) {
                            la = shA[(k) * kBlock + threadIdx.x];
                            lb = shB[(k) * kBlock + threadIdx.x];
                        } else {
                            load_lmk(idf[k], slot_of(k), buf_of(k), la, lb);
                        }
                        first_pass(k, la, lb);
                    }
                }
                // sample from the proposal (:353) ; weight terms (:360-367)
                const L3 Lp = llt3(P[0], P[3], P[4], P[6], P[7], P[8]);
                float xs = x, ys = y, ths = th;
                if constexpr (PP) {
                    if (pp_any != 0) mvgauss3(xs, ys, ths, Lp, g0, g1, g2);  // (a particle this step does not concern is not sampled)
                } else {
                    mvgauss3(xs, ys, ths, Lp, g0, g1, g2);
                }
                const float a0 = x0 - xs, a1 = y0 - ys, a2 = trig_offset(th0 - ths);
                const float b0 = x - xs, b1 = y - ys, b2 = trig_offset(th - ths);
                float lik = 1.0f;
                double dl = 0.0;  // log-weight contexts: sum of gaussEvaluate(.., logflag = 1)
                auto second_pass = [&](int k, float4 la, float lb) {
                    [[maybe_unused]] int jo = 0;
                    if constexpr (PP) {
                        jo = pp_j(k);
                        if (jo < 0 || la.x != la.x) {  // (not this particle's landmark in this step: the record moves on unchanged)
                            store_lmk(idf[k], buf_of(k), la, lb);
                            return;
                        }
                    }
                    Jac j = jacobian(xs, ys, ths, la.x, la.y, la.z, la.w, lb, r00, r01, r10, r11);
                    const float v0 = (PP ? pp_z(2 * jo) : zf[2 * k]) - j.zp0;
                    const float v1 = trig_offset((PP ? pp_z(2 * jo + 1) : zf[2 * k + 1]) - j.zp1);
                    if (logw) dl += (double) gauss2_log(v0, v1, j.s00, j.s10, j.s11);
                    else lik = lik * gauss2(v0, v1, j.s00, j.s10, j.s11);
                    cholesky_update2(la.x, la.y, la.z, la.w, lb, v0, v1, r00, r01, r10, r11, j.hf00, j.hf01, j.hf10, j.hf11);
                    store_lmk(idf[k], buf_of(k), la, lb);
                };
                // two loops on purpose: the LDS-fed one issues only stores to HBM, so nothing in it has to wait for a
                // store to land (a global load after a global store costs an s_waitcnt vmcnt(0) per iteration)
                if constexpr (BIG) {
                    pipeline(second_pass);
                } else {
                    const int ms = min(m, kStage);
                    for (int k = 0; k < ms; k++) second_pass(k, shA[(k) * kBlock + threadIdx.x], shB[(k) * kBlock + threadIdx.x]);
                    for (int k = ms; k < m; k++) {
                        float4 la;
                        float lb;
                        load_lmk(idf[k], slot_of(k), buf_of(k), la, lb);
                        second_pass(k, la, lb);
                    }
                }
                // (PP: no landmark matched: the proposal IS the prior, the two Gaussians cancel; nothing at all: pose and Pv stay)
                const bool weigh = !PP || (pp_any & 1) != 0, moved = !PP || pp_any != 0;
                if (!weigh) {
                } else if (logw) {
                    const float prior = gauss3_log(a0, a1, a2, q00, q10, q11, q20, q21, q22);
                    const float prop = gauss3_log(b0, b1, b2, P[0], P[3], P[4], P[6], P[7], P[8]);
                    w = (float) (((double) w + dl) + ((double) prior - (double) prop));
                } else {
                    const float prior = gauss3(a0, a1, a2, q00, q10, q11, q20, q21, q22);
                    const float prop = gauss3(b0, b1, b2, P[0], P[3], P[4], P[6], P[7], P[8]);
                    w = w * lik * prior / prop;
                }
                if (moved) {
                    x = xs;
                    y = ys;
                    th = ths;
                    q00 = q10 = q11 = q20 = q21 = q22 = 0.0f;
                    pose_dirty = true;
                }
            } else if (n > 0 && (!PP || pp_any != 0)) {
                // no re-observed landmark: sample the pose from the predicted Gaussian (fastslam2.cpp:36-42)
                mvgauss3(x, y, th, llt3(q00, q10, q11, q20, q21, q22), g0, g1, g2);
                q00 = q10 = q11 = q20 = q21 = q22 = 0.0f;
                pose_dirty = true;
            }
        } else {
            // FastSLAM 1: computeWeight (fastslam1.cpp:91-118) + featureUpdate at the particle pose
            if (m > 0) {
                float wp = 1.0f;
                double dl = 0.0;
#ifdef SLAM_FAST_MATH
                // fast build (round 4): the restructured arithmetic FastSLAM2's second pass uses (device_math.h: observe2 +
                // feature_update2: closed-form 2x2 inverse on one v_rcp_f32, polynomial atan2, hardware exp): computeWeight's
                // factor exp(-v^T S^-1 v / 2) / (2 pi sqrt(det S)) (fastslam1.cpp:105-115) is gaussEvaluate(v, S) with D = 2, and
                // choleskyUpdate is the same Kalman update.  ~100 VALU instructions per landmark instead of ~450 (IEEE divisions,
                // libm atan2f / expf / sqrtf): 1.2 -> ~0.3 us per landmark for a wave that has its SIMD to itself.
                const float rl1 = 0.5f * (r01 + r10);
                // (PP: the same operations on the particle's own observation; the other instantiations keep their text to the letter --
                // routing their z through a pair of temporaries changed the register allocation of the persistent loop)
                auto one_pass = [&](int k, float4 la, float lb) {
                    if constexpr (PP) {
                        const int j = pp_j(k);
                        if (j >= 0 && la.x == la.x) {
                            const Obs2 o = observe2(x, y, th, la.x, la.y, la.z, la.w, lb, r00, rl1, r11);
                            const Gauss2 g = feature_update2(la.x, la.y, la.z, la.w, lb, o, pp_z(2 * j) - o.zp0, wrap_pi(pp_z(2 * j + 1) - o.zp1));
                            if (logw) dl += (double) (g.E + __logf(g.norm));
                            else wp *= __expf(g.E) * g.norm;
                        }  // (else: not this particle's landmark in this step: the record moves on unchanged)
                        store_lmk(idf[k], buf_of(k), la, lb);
                    } else {
                        const Obs2 o = observe2(x, y, th, la.x, la.y, la.z, la.w, lb, r00, rl1, r11);
                        const Gauss2 g = feature_update2(la.x, la.y, la.z, la.w, lb, o, zf[2 * k] - o.zp0, wrap_pi(zf[2 * k + 1] - o.zp1));
                        if (logw) dl += (double) (g.E + __logf(g.norm));
                        else wp *= __expf(g.E) * g.norm;
                        store_lmk(idf[k], buf_of(k), la, lb);
                    }
                };
                // landmarks k and k + 1 in one basic block: the two updates are independent (only the weight product runs
                // through both, in landmark order), and a wave that has its SIMD to itself issues a DEPENDENT instruction
                // every ~9.6 cycles but two interleaved chains at 5.6 each (tools/microbench/valu_latency.hip): same
                // operations on the same values, bit-identical results
                auto pair_pass = [&](int k, float4 la, float lb, float4 ma, float mb) {
                    const Obs2 o = observe2(x, y, th, la.x, la.y, la.z, la.w, lb, r00, rl1, r11);
                    const Obs2 p = observe2(x, y, th, ma.x, ma.y, ma.z, ma.w, mb, r00, rl1, r11);
                    const Gauss2 g = feature_update2(la.x, la.y, la.z, la.w, lb, o, zf[2 * k] - o.zp0, wrap_pi(zf[2 * k + 1] - o.zp1));
                    const Gauss2 h = feature_update2(ma.x, ma.y, ma.z, ma.w, mb, p, zf[2 * k + 2] - p.zp0, wrap_pi(zf[2 * k + 3] - p.zp1));
                    if (logw) {
                        dl += (double) (g.E + __logf(g.norm));
                        dl += (double) (h.E + __logf(h.norm));
                    } else {
                        wp *= __expf(g.E) * g.norm;
                        wp *= __expf(h.E) * h.norm;
                    }
                    store_lmk(idf[k], buf_of(k), la, lb);
                    store_lmk(idf[k + 1], buf_of(k + 1), ma, mb);
                };
#else
                auto one_pass = [&](int k, float4 la, float lb) {
                    [[maybe_unused]] int jo = 0;
                    if constexpr (PP) {
                        jo = pp_j(k);
                        if (jo < 0 || la.x != la.x) {  // (not this particle's landmark in this step: the record moves on unchanged)
                            store_lmk(idf[k], buf_of(k), la, lb);
                            return;
                        }
                    }
                    Jac j = jacobian(x, y, th, la.x, la.y, la.z, la.w, lb, r00, r01, r10, r11);
                    const float v0 = (PP ? pp_z(2 * jo) : zf[2 * k]) - j.zp0;
                    const float v1 = trig_offset((PP ? pp_z(2 * jo + 1) : zf[2 * k + 1]) - j.zp1);
                    const float den = (float) (2 * kPi * (double) sqrtf(determinant2(j.s00, j.s01, j.s10, j.s11)));
                    float i00, i01, i10, i11;
                    inverse2(j.s00, j.s01, j.s10, j.s11, i00, i01, i10, i11);
                    const float t0 = -0.5f * (v0 * i00 + v1 * i10);
                    const float t1 = -0.5f * (v0 * i01 + v1 * i11);
                    if (logw) {
                        dl += (double) ((t0 * v0 + t1 * v1) - logf(den));
                    } else {
                        const float num = expf(t0 * v0 + t1 * v1);
                        wp = wp * num / den;
                    }
                    cholesky_update2(la.x, la.y, la.z, la.w, lb, v0, v1, r00, r01, r10, r11, j.hf00, j.hf01, j.hf10, j.hf11);
                    store_lmk(idf[k], buf_of(k), la, lb);
                };
#endif
                if constexpr (BIG) {
                    pipeline(one_pass);
                } else {
                    // staged like FastSLAM2's (round 4): the slots were requested with the pose and the records one trip behind
                    // them, all in flight together, instead of a slot -> record chain of two dependent trips per landmark
                    // (config 2: ~9 of the launch's 15 us lay between the arrival of the pose and its store)
                    stage_landmarks(ts, early_records);
                    SLAM_STAMP(5);  // records staged
                    int k = 0;
#ifdef SLAM_FAST_MATH
                    for (; k + 1 < min(m, kStage); k += 2)
                        pair_pass(k, shA[(k) * kBlock + threadIdx.x], shB[(k) * kBlock + threadIdx.x], shA[(k + 1) * kBlock + threadIdx.x],
                                  shB[(k + 1) * kBlock + threadIdx.x]);
#endif
                    for (; k < m; k++) {
                        float4 la;
                        float lb;
                        if (k < kStage) {
                            la = shA[(k) * kBlock + threadIdx.x];
                            lb = shB[(k) * kBlock + threadIdx.x];
                        } else {
                            load_lmk(idf[k], slot_of(k), buf_of(k), la, lb);
                        }
                        one_pass(k, la, lb);
                    }
                    SLAM_STAMP(7);  // landmark pass done, record stores landed
                }
                w = logw ? (float) ((double) w + dl) : w * wp;
            }
        }
        // addFeature (core.cpp:479-509): new landmarks appended at nf, nf+1, ...
#ifdef SLAM_FAST_MATH
        if (METHOD != 2)  // the fast FastSLAM2 branch above has already added them
#endif
        for (int k = 0; k < n; k++) {
            float4 la;
            float lb;
            if constexpr (PP) {
                const int j = pp_j(m + k);
                if (j >= 0) {
                    add_feature(x, y, th, pp_z(2 * j), pp_z(2 * j + 1), r00, r01, r10, r11, la.x, la.y, la.z, la.w, lb);
                } else {  // the particle does not open this landmark: an absent record (kernels.h: kAbsent)
                    la = make_float4(kAbsent, kAbsent, 0.0f, 0.0f);
                    lb = 0.0f;
                }
                store_new(ppa.idn[k], la, lb);
            } else {
                add_feature(x, y, th, zn[2 * k], zn[2 * k + 1], r00, r01, r10, r11, la.x, la.y, la.z, la.w, lb);
                store_new(nf + k, la, lb);
            }
        }
        // Row consolidation (slamgpu.cpp: do_update): landmarks out of view whose rows have gone stale are rewritten, unchanged,
        // into this particle's own slot of the row's other buffer and join the row this update opens -- one 40-byte move per
        // particle and landmark, once, instead of 4 bytes per particle, row and resample for the rest of the run (compact
        // contexts: the genealogy composition is ~1.8 us of a 16 us step at 10^5 particles when 25 rows are alive; big maps: a
        // row goes stale every step, and without this a resample's copy grows with the length of the run)
        if constexpr (!BIG) {
            for (int c = 0; c < n_cons; c++) {
                float4 la;
                float lb;
                load_lmk(idf[m + c], slot_of(m + c), buf_of(m + c), la, lb);
                store_lmk(idf[m + c], buf_of(m + c), la, lb);
            }
        } else {
            // (four at a time: slots, then records, in flight together; named values, not arrays: a register array that is
            // written under a condition is demoted to scratch)
            for (int c0 = 0; c0 < n_cons; c0 += 4) {
                const int k0 = m + c0, k1 = m + min(c0 + 1, n_cons - 1), k2 = m + min(c0 + 2, n_cons - 1), k3 = m + min(c0 + 3, n_cons - 1);
                const int s0 = slot_of(k0), s1 = slot_of(k1), s2 = slot_of(k2), s3 = slot_of(k3);
                float4 a0, a1, a2, a3;
                float b0, b1, b2, b3;
                load_lmk(idf[k0], s0, buf_of(k0), a0, b0);
                load_lmk(idf[k1], s1, buf_of(k1), a1, b1);
                load_lmk(idf[k2], s2, buf_of(k2), a2, b2);
                load_lmk(idf[k3], s3, buf_of(k3), a3, b3);
                // (past the end the indices repeat the last landmark: the same record stored again, by the same thread)
                store_lmk(idf[k0], buf_of(k0), a0, b0);
                store_lmk(idf[k1], buf_of(k1), a1, b1);
                store_lmk(idf[k2], buf_of(k2), a2, b2);
                store_lmk(idf[k3], buf_of(k3), a3, b3);
            }
        }
        // the landmarks this update wrote are in this particle's own slot now: that is what the genealogy row this update
        // opens says for all of them (the copy roles of a pending gather compose the other rows)
        if (!BIG && copy_inline) {
            int4 *__restrict__ o4 = reinterpret_cast<int4 *>(genO);
#pragma unroll
            for (int c = 0; c < kChunks; c++)
                if (c < nchunks) {
                    int4 q = gq[c];
                    if (c == (e_new >> 2)) {  // (e_new = -1: never)
                        const int comp = e_new & 3, own = B.first + i;
                        if (comp == 0) q.x = own;
                        else if (comp == 1) q.y = own;
                        else if (comp == 2) q.z = own;
                        else q.w = own;
                    }
                    __builtin_nontemporal_store((int __attribute__((ext_vector_type(4)))){q.x, q.y, q.z, q.w},
                                                reinterpret_cast<int __attribute__((ext_vector_type(4))) *>(&o4[(size_t) c * S + i]));
                }
        } else if (e_new >= 0) {
            genO[gen_index(!BIG, S, e_new, (size_t) i)] = B.first + i;
        }
        if constexpr (PP) w = logw ? w + ppa.wf[i] : w * ppa.wf[i];  // the observations this particle leaves unexplained (PerParticle::wf)
        nt_store(&poseAo[i], make_float4(x, y, th, w));
        if (METHOD == 2 && pose_dirty) {
            nt_store(&poseBo[i], make_float4(q00, q10, q11, q20));
            __builtin_nontemporal_store(q21, &poseCo[i].x);
            __builtin_nontemporal_store(q22, &poseCo[i].y);
        }
    }

    SLAM_STAMP(8);  // pose / genealogy stores landed
    // the block's term of the previous step's estimate: the waves' parts now, thread 0's combination behind the barrier the
    // weight prefix needs anyway (linear weights: one barrier at the end of the launch instead of two)
    if (STEP_PLAN) ei_prev = wave_reduce_est<!DIST>(ei_prev, sh_est);
    auto est_out = [&]() {
        if (STEP_PLAN && threadIdx.x == 0) {
            const EstItem e = combine_waves_est(ei_prev, sh_est);
            double *p = ws.est_part[STEP_WPAR ^ 1] + (size_t) bt * 4;
            p[0] = e.sx;
            p[1] = e.sy;
            p[2] = (double) e.th;
            p[3] = (double) e.w;
        }
    };
    // log-weight contexts: the prefix / totals below are those of exp(l - M_b), M_b = the block's largest log-weight,
    // which travels as a third row of the totals (scan_block_totals rescales by exp(M_b - M))
    if (logw) {
        float mb = w;
#pragma unroll
        for (int d = kWave / 2; d > 0; d >>= 1) mb = fmaxf(mb, __shfl_xor(mb, d, kWave));
        if (lane == 0) sh_w[wv] = mb;
        __syncthreads();
        mb = fmaxf(fmaxf(sh_w[0], sh_w[1]), fmaxf(sh_w[2], sh_w[3]));
        __syncthreads();
        if (threadIdx.x == 0) ws.blk_w[STEP_WPAR][2 * ws.nblocks + bt] = mb;
        w = (w == -INFINITY) ? 0.0f : expf(w - mb);  // NaN log-weights stay NaN and are flagged by the plan (status)
    }
    // in-block inclusive prefix of w; block totals of w and of w^2 (fixed association: deterministic).  The sum of squares
    // is kept SCALE-FREE, as q = sum (w_i / T)^2 with T the block total (every partial a ratio <= 1): the reference computes
    // Neff from the normalised weights (core.cpp:784-788) and survives weights whose square overflows float32 (w > 1.8e19:
    // a dozen landmarks in one update); scan_block_totals rebuilds sum w^2 = q T^2 in double.
    const float s = wave_scan_f(w);
    const float tw = wave_last_f(s);  // this wave's total
    const float rw = tw > 0.0f ? w / tw : 0.0f;
    const float s2 = wave_sum_f(rw * rw);
    if (lane == kWave - 1) {
        sh_w[wv] = s;
        sh_w2[wv] = s2;
    }
    __syncthreads();
    est_out();
    float base = 0.0f;
#pragma unroll
    for (int k = 0; k < kBlock / kWave; k++)
        if (k < wv) base += sh_w[k];
    __builtin_nontemporal_store(base + s, &ws.lcum[STEP_WPAR][i]);
    if (threadIdx.x == kBlock - 1) {
        const float T = base + s;
        float q = 0.0f;
        if (T > 0.0f) {
#pragma unroll
            for (int k = 0; k < kBlock / kWave; k++) {
                const float f = sh_w[k] / T;
                q += sh_w2[k] * (f * f);
            }
        }
        ws.blk_w[STEP_WPAR][bt] = T;
        ws.blk_w[STEP_WPAR][ws.nblocks + bt] = q;
        if (DIST && U.push_totals) {
            // push collective: this block's totals straight into every shard's table, shard-major [shard][w(nb) | q(nb)]
            // (visible to the peers' next launch: the flag handshake that follows this launch orders them)
            const size_t at = (size_t) B.shard * 2 * ws.nblocks + bt;
            for (int h = 0; h < B.n_shards; h++) {
                float *g = B.peers[h].gtot[STEP_WPAR];
                g[at] = T;
                g[at + ws.nblocks] = q;
            }
        }
    }
    SLAM_STAMP(9);  // weight prefix + totals written: end of the block
