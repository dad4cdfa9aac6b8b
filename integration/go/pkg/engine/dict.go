// dict.go — the id and row tables the shim keeps beside the engine: the C-ABI compares interned ids, never strings
// (include/kt_snapshot.h), and addresses pods / throttles / namespaces by ROW.  The Python test host does the same translation in
// kube_throttler_amd/objects.py (ClusterState) and the C++ plugin mirror in kube_throttler_amd/host/kt_host.cpp (Dict): this
// file is their Go statement.  NOT COMPILED HERE (no Go toolchain in the build image) — see engine.go.
package engine

import (
	"fmt"
	"sync"
)

// Dict interns label keys, (key, value) pairs, namespaces and resource names, and hands out rows with a free list.
type Dict struct {
	mu         sync.Mutex
	keys       map[string]uint32
	pairs      map[[2]string]uint32
	namespaces map[string]uint32
	dims       map[string]int // resource name -> dimension (<= 16 per engine; more names: one engine per page, kt_paged_*)
	scale      []int32        // decimal scale of every dimension: the smallest at which every quantity fed so far is an integer
	podRows    rowTable
	thrRows    rowTable
}

type rowTable struct {
	of   map[string]int64
	name []string
	free []int64
}

func (t *rowTable) get(key string) int64 {
	if t.of == nil {
		t.of = map[string]int64{}
	}
	if r, ok := t.of[key]; ok {
		return r
	}
	var r int64
	if n := len(t.free); n > 0 {
		r, t.free = t.free[n-1], t.free[:n-1]
		t.name[r] = key
	} else {
		r = int64(len(t.name))
		t.name = append(t.name, key)
	}
	t.of[key] = r
	return r
}

func (t *rowTable) release(key string) (int64, bool) {
	r, ok := t.of[key]
	if ok {
		delete(t.of, key)
		t.free = append(t.free, r)
	}
	return r, ok
}

// NewDict starts with ids 1.. (0 = "no label": kt_snapshot.h).
func NewDict() *Dict {
	return &Dict{keys: map[string]uint32{}, pairs: map[[2]string]uint32{}, namespaces: map[string]uint32{}, dims: map[string]int{}}
}

// Key / Pair intern a label key and a (key, value) pair.
func (d *Dict) Key(k string) uint32 {
	d.mu.Lock()
	defer d.mu.Unlock()
	if id, ok := d.keys[k]; ok {
		return id
	}
	id := uint32(len(d.keys) + 1)
	d.keys[k] = id
	return id
}
func (d *Dict) Pair(k, v string) uint32 {
	d.mu.Lock()
	defer d.mu.Unlock()
	kv := [2]string{k, v}
	if id, ok := d.pairs[kv]; ok {
		return id
	}
	id := uint32(len(d.pairs) + 1)
	d.pairs[kv] = id
	return id
}

// Namespace is the namespace ROW (rows are dense from 0: the engine sizes its per-namespace tables by the rows in use).
func (d *Dict) Namespace(name string) uint32 {
	d.mu.Lock()
	defer d.mu.Unlock()
	if id, ok := d.namespaces[name]; ok {
		return id
	}
	id := uint32(len(d.namespaces))
	d.namespaces[name] = id
	return id
}

// Dim is the dimension of a resource name; an engine holds at most 16.
func (d *Dict) Dim(name string) (int, error) {
	d.mu.Lock()
	defer d.mu.Unlock()
	if i, ok := d.dims[name]; ok {
		return i, nil
	}
	if len(d.dims) >= 16 {
		return 0, fmt.Errorf("resource name %q is the 17th of this engine: run one engine per page of 16 names (kt_paged_check / kt_paged_reconcile)", name)
	}
	i := len(d.dims)
	d.dims[name] = i
	d.scale = append(d.scale, 0)
	return i, nil
}

// PodRow / ThrottleRow return the row of an object, assigning one on first use; Release* recycles it on a Delete event.
func (d *Dict) PodRow(namespace, name string) int64 {
	d.mu.Lock()
	defer d.mu.Unlock()
	return d.podRows.get(namespace + "/" + name)
}
func (d *Dict) ThrottleRow(namespace, name string) int32 {
	d.mu.Lock()
	defer d.mu.Unlock()
	return int32(d.thrRows.get(namespace + "/" + name)) // ClusterThrottles: namespace ""
}
func (d *Dict) ReleasePod(namespace, name string) (int64, bool) {
	d.mu.Lock()
	defer d.mu.Unlock()
	return d.podRows.release(namespace + "/" + name)
}
func (d *Dict) ReleaseThrottle(namespace, name string) (int32, bool) {
	d.mu.Lock()
	defer d.mu.Unlock()
	r, ok := d.thrRows.release(namespace + "/" + name)
	return int32(r), ok
}

// ThrottleName is the NamespacedName behind a row of the status matrix (reason strings keep the reference's order because
// the shim sorts by it exactly where plugin.go:177-214 does).
func (d *Dict) ThrottleName(row int32) string {
	d.mu.Lock()
	defer d.mu.Unlock()
	if int(row) < len(d.thrRows.name) {
		return d.thrRows.name[row]
	}
	return ""
}
